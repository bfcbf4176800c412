// mgx_cube.cpp -- host-side structured-cube discretisation (see include/mgx_cube.h): what deal.II's
// Triangulation/DoFHandler/MatrixFree/FE_Q hand to the reference's LaplaceOperator and
// MultigridSolver for poisson_cube.  Pure host code (C++17 + OpenMP); the device work is behind
// mgx.h.
#include "../../include/mgx_cube.h"

#include <omp.h>
#include <sched.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <array>
#include <string>
#include <unordered_map>
#include <vector>

namespace
{
  constexpr int    kMaxN = 10;
  constexpr double kPi   = 3.14159265358979323846264338327950288;

  // ---- 1D element data: FE_Q(p) on Gauss-Lobatto nodes, QGauss(p+1) ----
  struct Basis
  {
    int    p = 0, n = 0;
    double gll[kMaxN], gq[kMaxN], gw[kMaxN];
    double S[kMaxN * kMaxN], D[kMaxN * kMaxN], P1[2 * kMaxN * kMaxN];
  };

  using ld = long double;

  void legendre(int n, ld x, ld &P, ld &dP)
  {
    if (n == 0)
      {
        P  = 1;
        dP = 0;
        return;
      }
    ld p0 = 1, p1 = x;
    for (int k = 2; k <= n; ++k)
      {
        const ld pk = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k;
        p0          = p1;
        p1          = pk;
      }
    P  = p1;
    dP = n * (x * p1 - p0) / (x * x - 1);
  }

  ld lagrange_value(const std::vector<ld> &x, int i, ld t)
  {
    ld v = 1;
    for (size_t j = 0; j < x.size(); ++j)
      if ((int)j != i)
        v *= (t - x[j]) / (x[i] - x[j]);
    return v;
  }

  ld lagrange_derivative(const std::vector<ld> &x, int i, ld t)
  {
    ld s = 0;
    for (size_t k = 0; k < x.size(); ++k)
      if ((int)k != i)
        {
          ld term = 1 / (x[i] - x[k]);
          for (size_t j = 0; j < x.size(); ++j)
            if ((int)j != i && j != k)
              term *= (t - x[j]) / (x[i] - x[j]);
          s += term;
        }
    return s;
  }

  void make_basis(Basis &b, int p)
  {
    const int n = p + 1;
    b.p         = p;
    b.n         = n;
    std::vector<ld> gll(n), gq(n), gw(n);
    for (int i = 0; i < n; ++i) // Gauss-Legendre nodes: Newton on P_n
      {
        ld x = -std::cos((ld)kPi * (i + 0.75L) / (n + 0.5L)), P, dP;
        for (int it = 0; it < 100; ++it)
          {
            legendre(n, x, P, dP);
            const ld dx = P / dP;
            x -= dx;
            if (std::fabs(dx) < 1e-19L)
              break;
          }
        legendre(n, x, P, dP);
        gq[i] = (x + 1) / 2;
        gw[i] = 1 / ((1 - x * x) * dP * dP);
      }
    gll[0] = 0;
    gll[p] = 1;
    for (int i = 1; i < p; ++i) // Gauss-Lobatto interior nodes: Newton on P_p'
      {
        ld x = -std::cos((ld)kPi * i / p), P, dP;
        for (int it = 0; it < 100; ++it)
          {
            legendre(p, x, P, dP);
            const ld d2P = (2 * x * dP - (ld)p * (p + 1) * P) / (1 - x * x);
            const ld dx  = dP / d2P;
            x -= dx;
            if (std::fabs(dx) < 1e-19L)
              break;
          }
        gll[i] = (x + 1) / 2;
      }
    for (int i = 0; i < n; ++i)
      {
        b.gll[i] = (double)gll[i];
        b.gq[i]  = (double)gq[i];
        b.gw[i]  = (double)gw[i];
      }
    for (int q = 0; q < n; ++q)
      for (int i = 0; i < n; ++i)
        {
          b.S[q * n + i] = (double)lagrange_value(gll, i, gq[q]);
          b.D[q * n + i] = (double)lagrange_derivative(gq, i, gq[q]);
        }
    for (int a = 0; a <= 2 * p; ++a)
      {
        const int child = a < p ? 0 : 1;
        const ld  xi    = (child + gll[a - child * p]) / 2;
        for (int i = 0; i < n; ++i)
          {
            ld v = lagrange_value(gll, i, xi);
            if (std::fabs(v) < 1e-18L)
              v = 0;
            if (std::fabs(v - 1) < 1e-18L)
              v = 1;
            b.P1[a * n + i] = (double)v;
          }
      }
  }

  struct Neighbor
  {
    int                   rank = -1;
    std::vector<uint32_t> index; // local DoFs shared with that rank, ordered by global grid id
  };

  struct Level
  {
    int                   level = 0;
    uint32_t              N = 0;          // local cells in x (all directions for a cube)
    uint32_t              Nl[3] = {0, 0, 0}, Ng[3] = {0, 0, 0}, off[3] = {0, 0, 0}; // local/global cells, offset
    uint32_t              n_cells = 0, n_dofs = 0, n_free = 0;
    double                h = 0;
    std::vector<uint32_t> idx27, idx27_plain, constrained, children, coords, dof_grid;
    std::vector<uint8_t>  brick_colour; // parity colouring of the 4x4x4 Morton bricks (level >= 2)
    std::vector<uint8_t>  weight_shift; // level >= 1: [n_parents*27] log2(global multiplicity) of patch entities
    std::vector<double>   rhs, bc_value;
    std::vector<uint32_t> bc_index;
    std::vector<Neighbor> neighbors;    // domain decomposition: interface exchange lists
    std::vector<uint32_t> shared, not_owned; // union of the lists; those owned by a lower rank
    // mapped mesh / variable coefficient (evaluate_coefficient's general branch): per cell and
    // quadrature point the merged coefficient, component-major [cell][6][n_q] with the weight
    // folded in, det(J) w_q and the physical quadrature point; empty on the Cartesian
    // constant-coefficient mesh
    uint32_t              cell_offset = 0; // multi-block meshes over ranks: index of local cell 0 among the cells of the whole level
    std::vector<double>   coef_q, jxw, xq;
    // multi-block meshes (hyper_shell): number of cells around each of the 27 entities of every cell
    // (the multiplicities of the transfer where they are not products of 1 and 2 per direction)
    std::vector<uint8_t>  ent_mult;
  };

  // one coarse cell of the hyper_shell meshes (mgx_shell.inc): the corner vectors c00, c10, c01, c11
  // (u, then v) of its face of the generating polyhedron
  struct ShellBlock
  {
    int c[4][3];
  };

  inline uint32_t compact3(uint32_t m)
  {
    uint32_t r = 0;
    for (int b = 0; b < 10; ++b)
      r |= ((m >> (3 * b)) & 1u) << b;
    return r;
  }

  // problem sets: MGX_CUBE_PROBLEM_CUBE poisson_cube/program.cc:98-104, 140-144, 266;
  // MGX_CUBE_PROBLEM_SHELL poisson_shell/program.cc:97-137 (solution), 157-200 (coefficient), 219-225 (rhs)
  struct Problem
  {
    int id = MGX_CUBE_PROBLEM_CUBE;
    double u(double x, double y, double z) const
    {
      if (id == MGX_CUBE_PROBLEM_SHELL)
        return std::sin(2. * kPi * (x + y));
      return f1(x) * f1(y) * f1(z);
    }
    // the cube problem's right-hand side is a product of one function per coordinate: f = f_scale f1(x) f1(y) f1(z)
    bool   f_is_product() const { return id != MGX_CUBE_PROBLEM_SHELL; }
    double f_scale() const { return 3. * kPi * 3. * kPi * 3.; }
    double f1(double x) const { return std::sin(kPi * x * 3.); }
    double a(const double *x) const
    {
      if (id != MGX_CUBE_PROBLEM_SHELL)
        return 1.;
      double prod = 1.;
      for (int e = 0; e < 3; ++e)
        {
          const double cs = std::cos(2. * kPi * x[e] + 0.1 * e);
          prod *= cs * cs;
        }
      return 1. + 1.0e6 * prod;
    }
    double f(double x, double y, double z) const
    {
      if (id != MGX_CUBE_PROBLEM_SHELL)
        return f_scale() * u(x, y, z);
      const double X[3]  = {x, y, z};
      const double arg   = 2. * kPi * (x + y);
      const double lap_u = -8. * kPi * kPi * std::sin(arg), du = 2. * kPi * std::cos(arg); // du/dx = du/dy
      double       grad_a_xy = 0.; // (grad a)_x + (grad a)_y: grad u has no z component
      for (int d = 0; d < 2; ++d)
        {
          double prod = 1.0e6;
          for (int e = 0; e < 3; ++e)
            {
              const double ang = 2. * kPi * X[e] + 0.1 * e, cs = std::cos(ang);
              prod *= (e == d) ? -4. * kPi * cs * std::sin(ang) : cs * cs;
            }
          grad_a_xy += prod;
        }
      return -(lap_u * a(X) + grad_a_xy * du);
    }
  };
} // namespace

struct mgx_cube_s
{
  int                p = 0;
  int                groots[3] = {1, 1, 1}; // coarse cells of the whole mesh
  int                procs[3]  = {1, 1, 1}; // process grid; every rank owns groots/procs coarse cells
  int                pcoord[3] = {0, 0, 0}, rank = 0, size = 1;
  int                lroots[3] = {1, 1, 1}, roff[3] = {0, 0, 0}; // this rank's coarse cells and offset
  double             origin = -0.9, h0 = 1.9;
  int                geometry = MGX_CUBE_GEOMETRY_CARTESIAN;
  Problem            problem;
  bool               brick_numbering = false;
  bool               mapped() const { return geometry != MGX_CUBE_GEOMETRY_CARTESIAN || problem.id != MGX_CUBE_PROBLEM_CUBE; }
  // reference box -> physical space
  void map(const double *X, double *x) const
  {
    if (geometry == MGX_CUBE_GEOMETRY_SHEARED)
      {
        x[0] = X[0] + 0.1 * (X[1] + 0.5 * X[2]);
        x[1] = X[1] + 0.1 * (0.3 * X[0] + X[2]);
        x[2] = X[2] + 0.1 * (0.2 * X[0] + 0.4 * X[1]);
      }
    else if (geometry == MGX_CUBE_GEOMETRY_SHELL_SECTOR)
      {
        // one block of hyper_shell(0, 0.5, 1, 6): equiangular cube-sphere sector around +z
        const double sx = (X[0] - origin) / (groots[0] * h0), sy = (X[1] - origin) / (groots[1] * h0),
                     sz = (X[2] - origin) / (groots[2] * h0);
        const double tx = std::tan((2. * sx - 1.) * kPi / 4.), ty = std::tan((2. * sy - 1.) * kPi / 4.);
        const double r  = (0.5 + 0.5 * sz) / std::sqrt(tx * tx + ty * ty + 1.);
        x[0]            = r * tx;
        x[1]            = r * ty;
        x[2]            = r;
      }
    else
      {
        x[0] = X[0];
        x[1] = X[1];
        x[2] = X[2];
      }
  }
  Basis              basis;
  std::vector<Level> levels;
  std::vector<ShellBlock> shell; // MGX_CUBE_GEOMETRY_HYPER_SHELL: the coarse cells of the whole mesh
  // ... over ranks: level i of this object is level i + level_offset of the whole mesh (1 where cells of level 1, not
  // coarse cells, are dealt out to the ranks: the coarser levels then exist on the undecomposed copy only)
  int                level_offset = 0;
};

namespace
{
  // Level of the (rank-local part of the) mesh.  Local cells in forest/Morton order over the
  // rank's coarse cells; all DoFs of the local cells get local indices (interface DoFs are
  // duplicated on the ranks that share them), Dirichlet DoFs of the GLOBAL boundary last.
  void build_level(mgx_cube_s &C, Level &L, int level)
  {
    const int p = C.p;
    L.level     = level;
    for (int d = 0; d < 3; ++d)
      {
        L.Nl[d]  = (uint32_t)C.lroots[d] << level;
        L.Ng[d]  = (uint32_t)C.groots[d] << level;
        L.off[d] = (uint32_t)C.roff[d] << level;
      }
    L.N               = L.Nl[0];
    L.n_cells         = L.Nl[0] * L.Nl[1] * L.Nl[2];
    L.h               = C.h0 / (double)(1u << level);
    const uint32_t nc = L.n_cells, per_root = 1u << (3 * level);
    const uint32_t rsx = (uint32_t)C.lroots[0], rsy = (uint32_t)C.lroots[1];
    L.coords.resize(3 * (size_t)nc);
    for (uint32_t c = 0; c < nc; ++c)
      {
        const uint32_t r = c / per_root, m = c % per_root;
        L.coords[3 * (size_t)c + 0] = ((r % rsx) << level) + compact3(m);
        L.coords[3 * (size_t)c + 1] = (((r / rsx) % rsy) << level) + compact3(m >> 1);
        L.coords[3 * (size_t)c + 2] = ((r / (rsx * rsy)) << level) + compact3(m >> 2);
      }
    const size_t          Ex = 2 * (size_t)L.Nl[0] + 1, Ey = 2 * (size_t)L.Nl[1] + 1, Ez = 2 * (size_t)L.Nl[2] + 1;
    const size_t          GEx = 2 * (size_t)L.Ng[0], GEy = 2 * (size_t)L.Ng[1], GEz = 2 * (size_t)L.Ng[2];
    std::vector<uint32_t> first(Ex * Ey * Ez, MGX_INVALID_INDEX);
    auto                  esize = [p](int cx, int cy, int cz) {
      return (uint32_t)((cx == 1 ? p - 1 : 1) * (cy == 1 ? p - 1 : 1) * (cz == 1 ? p - 1 : 1));
    };
    uint32_t next = 0;
    // Order of the numbering.  Cells are visited in Morton order and every mesh entity is numbered
    // by the first cell that contains it.  With `brick_numbering` the entities a brick of
    // NBd^3 consecutive cells numbers are additionally grouped by the brick-level entity they lie
    // on (brick interior, then its 2D faces, edges, corners): all DoFs that are complete after the
    // same set of colour launches of the device cell loop (mgx_brick.hip) then form long
    // contiguous runs, so no 128-B memory line mixes values that are finalised by different
    // launches.  The entity-contiguity contract of the compressed index table is untouched.
    const uint32_t nbd_num = p <= 4 ? 4 : 2, cb_num = nbd_num * nbd_num * nbd_num;
    const bool     grouped = C.brick_numbering && level >= (p <= 4 ? 2 : 1);
    struct Pending
    {
      uint32_t  key; // (group rank << 16) | visiting order
      uint32_t *slot;
      uint32_t  size;
    };
    std::vector<Pending> pending;
    for (int pass = 0; pass < 2; ++pass) // unconstrained entities first, Dirichlet boundary last
      {
        for (uint32_t c = 0; c < nc; ++c)
          {
            const size_t X = L.coords[3 * (size_t)c], Y = L.coords[3 * (size_t)c + 1],
                         Z = L.coords[3 * (size_t)c + 2];
            for (int cz = 0; cz < 3; ++cz)
              for (int cy = 0; cy < 3; ++cy)
                for (int cx = 0; cx < 3; ++cx)
                  {
                    const size_t ex = 2 * X + cx, ey = 2 * Y + cy, ez = 2 * Z + cz;
                    const size_t gex = 2 * (size_t)L.off[0] + ex, gey = 2 * (size_t)L.off[1] + ey,
                                 gez = 2 * (size_t)L.off[2] + ez;
                    const bool on_boundary =
                      gex == 0 || gex == GEx || gey == 0 || gey == GEy || gez == 0 || gez == GEz;
                    if ((int)on_boundary != pass)
                      continue;
                    uint32_t &f = first[(ez * Ey + ey) * Ex + ex];
                    if (f != MGX_INVALID_INDEX)
                      continue;
                    if (!grouped)
                      {
                        f = next;
                        next += esize(cx, cy, cz);
                        continue;
                      }
                    f = MGX_INVALID_INDEX - 1; // claimed, numbered when the brick is complete
                    // which brick surfaces the entity lies on: 0 interior, 1 low, 2 high per direction
                    const uint32_t E2 = 2 * nbd_num;
                    const uint32_t sx = ex % E2 == 0 ? ((X % nbd_num == 0 && cx == 0) ? 1 : 2) : 0;
                    const uint32_t sy = ey % E2 == 0 ? ((Y % nbd_num == 0 && cy == 0) ? 1 : 2) : 0;
                    const uint32_t sz = ez % E2 == 0 ? ((Z % nbd_num == 0 && cz == 0) ? 1 : 2) : 0;
                    const uint32_t nsurf = (sx != 0) + (sy != 0) + (sz != 0);
                    const uint32_t group = nsurf * 27 + sz * 9 + sy * 3 + sx; // interior, faces, edges, corners
                    pending.push_back({(group << 16) | (uint32_t)pending.size(), &f, esize(cx, cy, cz)});
                  }
            if (grouped && (c + 1) % cb_num == 0)
              {
                std::sort(pending.begin(), pending.end(), [](const Pending &a, const Pending &b) { return a.key < b.key; });
                for (const Pending &q : pending)
                  {
                    *q.slot = next;
                    next += q.size;
                  }
                pending.clear();
              }
          }
        if (pass == 0)
          L.n_free = next;
      }
    L.n_dofs = next;
    L.constrained.resize(L.n_dofs - L.n_free);
    for (uint32_t i = 0; i < L.constrained.size(); ++i)
      L.constrained[i] = L.n_free + i;
    L.idx27.resize(27 * (size_t)nc);
    L.idx27_plain.resize(27 * (size_t)nc);
#pragma omp parallel for schedule(static)
    for (uint32_t c = 0; c < nc; ++c)
      {
        const size_t X = L.coords[3 * (size_t)c], Y = L.coords[3 * (size_t)c + 1], Z = L.coords[3 * (size_t)c + 2];
        for (int e = 0; e < 27; ++e)
          {
            const int      cx = e % 3, cy = (e / 3) % 3, cz = e / 9;
            const uint32_t base = first[((2 * Z + cz) * Ey + 2 * Y + cy) * Ex + 2 * X + cx];
            L.idx27_plain[27 * (size_t)c + e] = base;
            L.idx27[27 * (size_t)c + e]       = base >= L.n_free ? MGX_INVALID_INDEX : base;
          }
      }
    // bricks of the device cell loop: 4x4x4 cells (64 consecutive Morton cells) for p <= 4,
    // 2x2x2 (the 8 children of a parent) for p >= 5; 8 colours by brick parity
    const uint32_t nbd = p <= 4 ? 4 : 2, cb = nbd * nbd * nbd;
    if (level >= (p <= 4 ? 2 : 1))
      {
        L.brick_colour.resize(nc / cb);
        for (uint32_t b = 0; b < nc / cb; ++b)
          {
            const uint32_t X = L.coords[3 * (size_t)(cb * b)] / nbd, Y = L.coords[3 * (size_t)(cb * b) + 1] / nbd,
                           Z = L.coords[3 * (size_t)(cb * b) + 2] / nbd;
            L.brick_colour[b] = (uint8_t)((X & 1) | ((Y & 1) << 1) | ((Z & 1) << 2));
          }
      }
    if (level > 0)
      {
        L.children.resize(nc);
        for (uint32_t i = 0; i < nc; ++i)
          L.children[i] = i; // Morton order: children of parent c are 8c .. 8c+7
        // 1/multiplicity weights of the transfer from the GLOBAL position of the
        // parent: a patch-boundary point is shared with the neighbouring parent's patch unless
        // it lies on the domain boundary -- also when that neighbour lives on another rank
        const Level   &Lc   = C.levels[level - 1];
        const uint32_t npar = nc / 8;
        L.weight_shift.resize(27 * (size_t)npar);
        for (uint32_t pc = 0; pc < npar; ++pc)
          {
            int sh[3][3];
            for (int d = 0; d < 3; ++d)
              {
                const uint32_t g = Lc.off[d] + Lc.coords[3 * (size_t)pc + d];
                sh[d][0]         = g > 0 ? 1 : 0;
                sh[d][1]         = 0;
                sh[d][2]         = g + 1 < Lc.Ng[d] ? 1 : 0;
              }
            for (int e = 0; e < 27; ++e)
              L.weight_shift[27 * (size_t)pc + e] = (uint8_t)(sh[0][e % 3] + sh[1][(e / 3) % 3] + sh[2][e / 9]);
          }
      }
  }

  // dof -> global lexicographic grid id, and (for a decomposed mesh) the interface lists
  void build_dof_grid(const mgx_cube_s &C, Level &L)
  {
    const int    p  = C.p;
    const size_t Gx = (size_t)L.Ng[0] * p + 1, Gy = (size_t)L.Ng[1] * p + 1;
    L.dof_grid.assign(L.n_dofs, 0);
    for (uint32_t c = 0; c < L.n_cells; ++c)
      {
        const size_t X = L.off[0] + L.coords[3 * (size_t)c], Y = L.off[1] + L.coords[3 * (size_t)c + 1],
                     Z = L.off[2] + L.coords[3 * (size_t)c + 2];
        for (int e = 0; e < 27; ++e)
          {
            const int      cx = e % 3, cy = (e / 3) % 3, cz = e / 9;
            const int      nx = cx == 1 ? p - 1 : 1, ny = cy == 1 ? p - 1 : 1, nz = cz == 1 ? p - 1 : 1;
            const uint32_t base = L.idx27_plain[27 * (size_t)c + e];
            for (int oz = 0; oz < nz; ++oz)
              for (int oy = 0; oy < ny; ++oy)
                for (int ox = 0; ox < nx; ++ox)
                  {
                    const size_t gx = X * p + (cx == 0 ? 0 : (cx == 2 ? p : 1 + ox));
                    const size_t gy = Y * p + (cy == 0 ? 0 : (cy == 2 ? p : 1 + oy));
                    const size_t gz = Z * p + (cz == 0 ? 0 : (cz == 2 ? p : 1 + oz));
                    L.dof_grid[base + (uint32_t)((oz * ny + oy) * nx + ox)] = (uint32_t)((gz * Gy + gy) * Gx + gx);
                  }
          }
      }
  }

  void build_interfaces(const mgx_cube_s &C, Level &L)
  {
    if (C.size == 1)
      return;
    const int    p  = C.p;
    const size_t Gx = (size_t)L.Ng[0] * p + 1, Gy = (size_t)L.Ng[1] * p + 1;
    // point range of this rank per direction
    size_t lo[3], hi[3];
    for (int d = 0; d < 3; ++d)
      {
        lo[d] = (size_t)L.off[d] * p;
        hi[d] = (size_t)(L.off[d] + L.Nl[d]) * p;
      }
    std::vector<std::pair<uint32_t, uint32_t>> lists[27]; // (global id, local index) per neighbour offset
    // Dirichlet DoFs are never written by the cell loop: they take no part in the exchange, but a
    // duplicated one must still be counted once in dot products
    std::vector<uint8_t> lower(L.n_dofs, 0), dup(L.n_dofs, 0);
    for (uint32_t i = 0; i < L.n_dofs; ++i)
      {
        const size_t gid = L.dof_grid[i];
        const size_t g[3] = {gid % Gx, (gid / Gx) % Gy, gid / (Gx * Gy)};
        int          lo_ok[3], hi_ok[3];
        bool         any = false;
        for (int d = 0; d < 3; ++d)
          {
            lo_ok[d] = g[d] == lo[d] && C.pcoord[d] > 0;
            hi_ok[d] = g[d] == hi[d] && C.pcoord[d] + 1 < C.procs[d];
            any      = any || lo_ok[d] || hi_ok[d];
          }
        if (!any)
          continue;
        dup[i] = 1;
        if (i < L.n_free)
          L.shared.push_back(i);
        for (int dz = -1; dz <= 1; ++dz)
          for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx)
              {
                if (!dx && !dy && !dz)
                  continue;
                const int dd[3] = {dx, dy, dz};
                bool      ok    = true;
                for (int d = 0; d < 3; ++d)
                  ok = ok && (dd[d] == 0 || (dd[d] < 0 ? lo_ok[d] : hi_ok[d]));
                if (!ok)
                  continue;
                if (i < L.n_free)
                  lists[(dz + 1) * 9 + (dy + 1) * 3 + dx + 1].push_back({(uint32_t)gid, i});
                const int nr =
                  ((C.pcoord[2] + dz) * C.procs[1] + C.pcoord[1] + dy) * C.procs[0] + C.pcoord[0] + dx;
                if (nr < C.rank)
                  lower[i] = 1;
              }
      }
    for (uint32_t i = 0; i < L.n_dofs; ++i)
      if (dup[i] && lower[i])
        L.not_owned.push_back(i);
    for (int k = 0; k < 27; ++k)
      {
        if (lists[k].empty())
          continue;
        const int dx = k % 3 - 1, dy = (k / 3) % 3 - 1, dz = k / 9 - 1;
        Neighbor  nb;
        nb.rank = ((C.pcoord[2] + dz) * C.procs[1] + C.pcoord[1] + dy) * C.procs[0] + C.pcoord[0] + dx;
        std::sort(lists[k].begin(), lists[k].end());
        nb.index.reserve(lists[k].size());
        for (auto &pr : lists[k])
          nb.index.push_back(pr.second);
        L.neighbors.push_back(std::move(nb));
      }
    std::sort(L.neighbors.begin(), L.neighbors.end(), [](const Neighbor &a, const Neighbor &b) { return a.rank < b.rank; });
  }

  // host cell kernels (setup only): lexicographic gather through a 27-entry table
  void gather_cell(int p, const uint32_t *base, const double *src, double *v)
  {
    const int n = p + 1;
    for (int k = 0; k < n; ++k)
      {
        const int cz = k == 0 ? 0 : (k == p ? 2 : 1), oz = cz == 1 ? k - 1 : 0;
        for (int j = 0; j < n; ++j)
          {
            const int       cy = j == 0 ? 0 : (j == p ? 2 : 1), oy = cy == 1 ? j - 1 : 0;
            const uint32_t  off = (uint32_t)((cy == 1 ? p - 1 : 1) * oz + oy);
            const uint32_t *ind = base + 3 * (3 * cz + cy);
            double         *row = v + (k * n + j) * n;
            row[0]              = ind[0] == MGX_INVALID_INDEX ? 0. : src[ind[0] + off];
            for (int i = 0; i < p - 1; ++i)
              row[1 + i] = ind[1] == MGX_INVALID_INDEX ? 0. : src[ind[1] + off * (uint32_t)(p - 1) + (uint32_t)i];
            row[p] = ind[2] == MGX_INVALID_INDEX ? 0. : src[ind[2] + off];
          }
      }
  }

  template <bool atomic>
  void scatter_cell(int p, const uint32_t *base, double *dst, const double *v)
  {
    const int n   = p + 1;
    auto      add = [&](uint32_t idx, double val) {
      if (atomic)
        {
#pragma omp atomic
          dst[idx] += val;
        }
      else
        dst[idx] += val;
    };
    for (int k = 0; k < n; ++k)
      {
        const int cz = k == 0 ? 0 : (k == p ? 2 : 1), oz = cz == 1 ? k - 1 : 0;
        for (int j = 0; j < n; ++j)
          {
            const int       cy = j == 0 ? 0 : (j == p ? 2 : 1), oy = cy == 1 ? j - 1 : 0;
            const uint32_t  off = (uint32_t)((cy == 1 ? p - 1 : 1) * oz + oy);
            const uint32_t *ind = base + 3 * (3 * cz + cy);
            const double   *row = v + (k * n + j) * n;
            if (ind[0] != MGX_INVALID_INDEX)
              add(ind[0] + off, row[0]);
            if (ind[1] != MGX_INVALID_INDEX)
              for (int i = 0; i < p - 1; ++i)
                add(ind[1] + off * (uint32_t)(p - 1) + (uint32_t)i, row[1 + i]);
            if (ind[2] != MGX_INVALID_INDEX)
              add(ind[2] + off, row[p]);
          }
      }
  }

  // tensor-product application of a 1D matrix (n x n, row-major, or its transpose) along dir
  void apply_1d(int n, int dir, const double *M, bool transpose, const double *in, double *out, bool add)
  {
    const int stride = dir == 0 ? 1 : (dir == 1 ? n : n * n);
    for (int o2 = 0; o2 < n; ++o2)
      for (int o1 = 0; o1 < n; ++o1)
        {
          const int base = dir == 0 ? (o2 * n + o1) * n : (dir == 1 ? o2 * n * n + o1 : o2 * n + o1);
          for (int a = 0; a < n; ++a)
            {
              double s = 0;
              for (int b = 0; b < n; ++b)
                s += (transpose ? M[b * n + a] : M[a * n + b]) * in[base + b * stride];
              if (add)
                out[base + a * stride] += s;
              else
                out[base + a * stride] = s;
            }
        }
  }

} // namespace
#include "mgx_shell.inc"
namespace
{
  // evaluate_coefficient, general branch (laplace_operator.h:388-430) for a mapped mesh: the cell
  // geometry is the degree-p interpolant of the map at the GLL support points (MappingQ of
  // multigrid_solver.h:139); per quadrature point JxW = det J w_q and coef = a(x_q) JxW J^-1 J^-T
  void build_geometry(const mgx_cube_s &C, Level &L)
  {
    const int    p = C.p, n = p + 1, n3 = n * n * n;
    const Basis &B = C.basis;
    L.coef_q.assign((size_t)L.n_cells * 6 * n3, 0.);
    L.jxw.assign((size_t)L.n_cells * n3, 0.);
    L.xq.assign((size_t)L.n_cells * n3 * 3, 0.);
#pragma omp parallel
    {
      std::vector<double> buf(14 * (size_t)n3);
      double *nodes = buf.data(), *val = nodes + 3 * n3, *tmp = val + n3, *dx = tmp + n3; // dx: 9 n3 (d x_a / d xi_b)
#pragma omp for schedule(static)
      for (uint32_t c = 0; c < L.n_cells; ++c)
        {
          const double X0[3] = {C.origin + L.h * (L.off[0] + L.coords[3 * (size_t)c]),
                                C.origin + L.h * (L.off[1] + L.coords[3 * (size_t)c + 1]),
                                C.origin + L.h * (L.off[2] + L.coords[3 * (size_t)c + 2])};
          if (C.geometry == MGX_CUBE_GEOMETRY_HYPER_SHELL)
            shell_cell_nodes(C, L, c, nodes);
          else
          for (int k = 0, i3 = 0; k < n; ++k)
            for (int j = 0; j < n; ++j)
              for (int i = 0; i < n; ++i, ++i3)
                {
                  const double Xr[3] = {X0[0] + L.h * B.gll[i], X0[1] + L.h * B.gll[j], X0[2] + L.h * B.gll[k]};
                  double       xp[3];
                  C.map(Xr, xp);
                  nodes[i3] = xp[0];
                  nodes[n3 + i3] = xp[1];
                  nodes[2 * n3 + i3] = xp[2];
                }
          for (int a = 0; a < 3; ++a)
            {
              apply_1d(n, 0, B.S, false, nodes + a * n3, val, false);
              apply_1d(n, 1, B.S, false, val, tmp, false);
              apply_1d(n, 2, B.S, false, tmp, val, false);
              for (int q = 0; q < n3; ++q)
                L.xq[((size_t)c * n3 + q) * 3 + a] = val[q];
              for (int b = 0; b < 3; ++b)
                apply_1d(n, b, B.D, false, val, dx + (3 * a + b) * n3, false);
            }
          for (int k = 0, q = 0; k < n; ++k)
            for (int j = 0; j < n; ++j)
              for (int i = 0; i < n; ++i, ++q)
                {
                  double F[9], inv[9]; // F[3a+b] = d x_a / d xi_b
                  for (int m = 0; m < 9; ++m)
                    F[m] = dx[m * n3 + q];
                  const double c00 = F[4] * F[8] - F[5] * F[7], c01 = F[5] * F[6] - F[3] * F[8], c02 = F[3] * F[7] - F[4] * F[6];
                  const double det = F[0] * c00 + F[1] * c01 + F[2] * c02;
                  // inverse by cofactors: inv[3b+a] = d xi_b / d x_a
                  inv[0] = c00 / det;
                  inv[3] = c01 / det;
                  inv[6] = c02 / det;
                  inv[1] = (F[2] * F[7] - F[1] * F[8]) / det;
                  inv[4] = (F[0] * F[8] - F[2] * F[6]) / det;
                  inv[7] = (F[1] * F[6] - F[0] * F[7]) / det;
                  inv[2] = (F[1] * F[5] - F[2] * F[4]) / det;
                  inv[5] = (F[2] * F[3] - F[0] * F[5]) / det;
                  inv[8] = (F[0] * F[4] - F[1] * F[3]) / det;
                  const double  jxw = det * B.gw[i] * B.gw[j] * B.gw[k];
                  const double *xp  = &L.xq[((size_t)c * n3 + q) * 3];
                  const double  s   = C.problem.a(xp) * jxw;
                  double       *Cq  = &L.coef_q[(size_t)c * 6 * n3];
                  auto          dotr = [&](int e, int f) { return inv[3 * e] * inv[3 * f] + inv[3 * e + 1] * inv[3 * f + 1] + inv[3 * e + 2] * inv[3 * f + 2]; };
                  Cq[q]          = s * dotr(0, 0);
                  Cq[n3 + q]     = s * dotr(1, 1);
                  Cq[2 * n3 + q] = s * dotr(2, 2);
                  Cq[3 * n3 + q] = s * dotr(0, 1);
                  Cq[4 * n3 + q] = s * dotr(0, 2);
                  Cq[5 * n3 + q] = s * dotr(1, 2);
                  L.jxw[(size_t)c * n3 + q] = jxw;
                }
        }
    }
  }

  // boundary values (multigrid_solver.h:225-253)
  void build_bc(const mgx_cube_s &C, Level &L)
  {
    const int    p = C.p;
    const Basis &B = C.basis;
    // inhomogeneous_bc: analytic solution at the support points of boundary DoFs, nonzero only
    std::vector<double> bc_full(L.n_dofs, 0.);
    if (C.geometry == MGX_CUBE_GEOMETRY_HYPER_SHELL)
      shell_boundary_values(C, L, bc_full);
    else
    {
      // support point coordinate of LOCAL grid index g along direction d
      std::vector<double> xd[3];
      for (int d = 0; d < 3; ++d)
        {
          const size_t G = (size_t)L.Nl[d] * p + 1;
          xd[d].resize(G);
          for (size_t g = 0; g < G; ++g)
            {
              size_t cell = g / p, loc = g % p;
              if (cell == L.Nl[d])
                {
                  cell = L.Nl[d] - 1;
                  loc  = p;
                }
              xd[d][g] = C.origin + L.h * ((double)(L.off[d] + cell) + B.gll[loc]);
            }
        }
      // walk boundary entities through the cells that touch the boundary
      std::vector<uint8_t> done(L.n_dofs - L.n_free, 0);
      for (uint32_t c = 0; c < L.n_cells; ++c)
        {
          const size_t X = L.coords[3 * (size_t)c], Y = L.coords[3 * (size_t)c + 1], Z = L.coords[3 * (size_t)c + 2];
          if (X != 0 && X != L.Nl[0] - 1 && Y != 0 && Y != L.Nl[1] - 1 && Z != 0 && Z != L.Nl[2] - 1)
            continue;
          for (int e = 0; e < 27; ++e)
            {
              const uint32_t base = L.idx27_plain[27 * (size_t)c + e];
              const int      cx = e % 3, cy = (e / 3) % 3, cz = e / 9;
              const int      nx = cx == 1 ? p - 1 : 1, ny = cy == 1 ? p - 1 : 1, nz = cz == 1 ? p - 1 : 1;
              if (nx * ny * nz == 0) // p = 1: lines/quads/hexes carry no DoFs
                continue;
              if (base < L.n_free || done[base - L.n_free])
                continue;
              done[base - L.n_free] = 1;
              for (int oz = 0; oz < nz; ++oz)
                for (int oy = 0; oy < ny; ++oy)
                  for (int ox = 0; ox < nx; ++ox)
                    {
                      const size_t gx = X * p + (cx == 0 ? 0 : (cx == 2 ? p : 1 + ox));
                      const size_t gy = Y * p + (cy == 0 ? 0 : (cy == 2 ? p : 1 + oy));
                      const size_t gz = Z * p + (cz == 0 ? 0 : (cz == 2 ? p : 1 + oz));
                      {
                        const double Xr[3] = {xd[0][gx], xd[1][gy], xd[2][gz]};
                        double       xp[3];
                        C.map(Xr, xp);
                        bc_full[base + (uint32_t)((oz * ny + oy) * nx + ox)] = C.problem.u(xp[0], xp[1], xp[2]);
                      }
                    }
            }
        }
    }
    for (uint32_t i = L.n_free; i < L.n_dofs; ++i)
      if (bc_full[i] != 0.0) // multigrid_solver.h:250
        {
          L.bc_index.push_back(i);
          L.bc_value.push_back(bc_full[i]);
        }
  }

  // f(x_q) JxW_q at the n^3 quadrature points of cell c (laplace_operator.h:839)
  void rhs_quadrature_cell(const mgx_cube_s &C, const Level &L, uint32_t c, double *t0)
  {
    const int    n = C.p + 1, n3 = n * n * n;
    const Basis &B = C.basis;
    if (!L.coef_q.empty())
      {
        for (int q = 0; q < n3; ++q)
          {
            const double *xp = &L.xq[((size_t)c * n3 + q) * 3];
            t0[q]            = C.problem.f(xp[0], xp[1], xp[2]) * L.jxw[(size_t)c * n3 + q];
          }
        return;
      }
    const double h = L.h, h3 = h * h * h;
    const double x0 = C.origin + h * (L.off[0] + L.coords[3 * (size_t)c]), y0 = C.origin + h * (L.off[1] + L.coords[3 * (size_t)c + 1]),
                 z0 = C.origin + h * (L.off[2] + L.coords[3 * (size_t)c + 2]);
    if (C.problem.f_is_product())
      {
        // 3 n evaluations of the 1D factor instead of n^3 of f; the products in the order of Problem::f and of the
        // general expression below (the values are the same to the last bit)
        double fx[16], fy[16], fz[16];
        for (int i = 0; i < n; ++i)
          {
            fx[i] = C.problem.f1(x0 + h * B.gq[i]);
            fy[i] = C.problem.f1(y0 + h * B.gq[i]);
            fz[i] = C.problem.f1(z0 + h * B.gq[i]);
          }
        for (int k = 0, q = 0; k < n; ++k)
          for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i, ++q)
              t0[q] = C.problem.f_scale() * (fx[i] * fy[j] * fz[k]) * h3 * (B.gw[i] * B.gw[j] * B.gw[k]);
        return;
      }
    for (int k = 0, q = 0; k < n; ++k)
      for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i, ++q)
          t0[q] = C.problem.f(x0 + h * B.gq[i], y0 + h * B.gq[j], z0 + h * B.gq[k]) * h3 * (B.gw[i] * B.gw[j] * B.gw[k]);
  }

  // rhs on the host (laplace_operator.h:804-845), built when first asked for (mgx_cube_rhs; a solver created with
  // device_rhs assembles it on the GPU, mgx_solver_compute_rhs)
  void build_rhs(const mgx_cube_s &C, Level &L)
  {
    const int    p = C.p, n = p + 1, n3 = n * n * n;
    const Basis &B = C.basis;
    std::vector<double> bc_full(L.n_dofs, 0.);
    for (size_t i = 0; i < L.bc_index.size(); ++i)
      bc_full[L.bc_index[i]] = L.bc_value[i];
    L.rhs.assign(L.n_dofs, 0.);
    const double h = L.h;
#pragma omp parallel
    {
      std::vector<double> buf(5 * (size_t)n3);
      double             *u = buf.data(), *t0 = u + n3, *gx = t0 + n3, *gy = gx + n3, *gz = gy + n3;
#pragma omp for schedule(static)
      for (uint32_t c = 0; c < L.n_cells; ++c)
        {
          gather_cell(p, &L.idx27_plain[27 * (size_t)c], bc_full.data(), u);
          for (int i = 0; i < n3; ++i)
            u[i] = -u[i]; // laplace_operator.h:823-824
          apply_1d(n, 0, B.S, false, u, t0, false);
          apply_1d(n, 1, B.S, false, t0, u, false);
          apply_1d(n, 2, B.S, false, u, t0, false);
          apply_1d(n, 0, B.D, false, t0, gx, false);
          apply_1d(n, 1, B.D, false, t0, gy, false);
          apply_1d(n, 2, B.D, false, t0, gz, false);
          if (!L.coef_q.empty()) // general branch: full tensor per quadrature point
            for (int q = 0; q < n3; ++q)
              {
                const double *Cq = &L.coef_q[(size_t)c * 6 * n3];
                const double  a = gx[q], b = gy[q], cc = gz[q];
                gx[q] = Cq[q] * a + Cq[3 * n3 + q] * b + Cq[4 * n3 + q] * cc;
                gy[q] = Cq[3 * n3 + q] * a + Cq[n3 + q] * b + Cq[5 * n3 + q] * cc;
                gz[q] = Cq[4 * n3 + q] * a + Cq[5 * n3 + q] * b + Cq[2 * n3 + q] * cc;
              }
          else
          for (int k = 0, q = 0; k < n; ++k)
            for (int j = 0; j < n; ++j)
              for (int i = 0; i < n; ++i, ++q)
                {
                  const double w = B.gw[i] * B.gw[j] * B.gw[k];
                  gx[q] *= h * w; // merged coefficient diag(h,h,h) times w_q
                  gy[q] *= h * w;
                  gz[q] *= h * w;
                }
          rhs_quadrature_cell(C, L, c, t0); // :839
          apply_1d(n, 0, B.D, true, gx, t0, true);
          apply_1d(n, 1, B.D, true, gy, t0, true);
          apply_1d(n, 2, B.D, true, gz, t0, true);
          apply_1d(n, 0, B.S, true, t0, u, false);
          apply_1d(n, 1, B.S, true, u, t0, false);
          apply_1d(n, 2, B.S, true, t0, u, false);
          scatter_cell<true>(p, &L.idx27[27 * (size_t)c], L.rhs.data(), u);
        }
    }
  }

  // threads the host really grants: min(affinity mask, cgroup CPU quota) unless OMP_NUM_THREADS
  // is set (a GPU box hands a small CPU share of a large machine to each job)
  int effective_threads()
  {
    const char *env = std::getenv("OMP_NUM_THREADS");
    if (env && std::atoi(env) > 0)
      return std::atoi(env);
    int       n = omp_get_num_procs();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0)
      n = CPU_COUNT(&set);
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r"))
      {
        char   quota[64];
        double period = 0;
        if (std::fscanf(f, "%63s %lf", quota, &period) == 2 && std::strcmp(quota, "max") != 0 && period > 0)
          {
            const int q = (int)(std::atof(quota) / period + 0.5);
            if (q >= 1 && q < n)
              n = q;
          }
        std::fclose(f);
      }
    return std::max(1, n);
  }
} // namespace

namespace mgx
{
  int report_error(int code, const char *message); // mgx_api.cpp: message for mgx_last_error()
}

extern "C" {

static int create_impl(const mgx_cube_box_desc &bd, mgx_cube_t *out)
{
  const int degree = bd.degree, n_refine = bd.n_refine;
  if (!out || degree < 1 || degree > MGX_MAX_DEGREE || n_refine < 0 || n_refine > 9)
    return mgx::report_error(MGX_ERR_INVALID_ARGUMENT, "mgx_cube_create: degree must be in 1..9 and n_refine in 0..9");
  uint64_t total = 1;
  int      size  = 1;
  for (int d = 0; d < 3; ++d)
    {
      if (bd.roots[d] < 1 || bd.procs[d] < 1 || bd.roots[d] % bd.procs[d] != 0)
        return mgx::report_error(MGX_ERR_INVALID_ARGUMENT, "mgx_cube_create: the process grid must divide the coarse cells");
      const uint64_t N = (uint64_t)bd.roots[d] << n_refine;
      if (N > 2047)
        return mgx::report_error(MGX_ERR_UNSUPPORTED, "mgx_cube_create: more than 2047 cells per direction");
      total *= N * degree + 1;
      size *= bd.procs[d];
    }
  if (total >= 0xFFFFFFFFull)
    return mgx::report_error(MGX_ERR_UNSUPPORTED, "mgx_cube_create: more than 2^32 global DoFs"); // 32-bit (global) DoF indices as in the reference's compressed table
  if (bd.rank < 0 || bd.rank >= size || !(bd.h0 > 0))
    return mgx::report_error(MGX_ERR_INVALID_ARGUMENT, "mgx_cube_create: rank outside the process grid or h0 <= 0");
  if (bd.numbering != MGX_CUBE_NUMBERING_BRICK && bd.numbering != MGX_CUBE_NUMBERING_CELL)
    return mgx::report_error(MGX_ERR_INVALID_ARGUMENT, "mgx_cube_create: unknown numbering");
  if (bd.geometry < 0 || bd.geometry > MGX_CUBE_GEOMETRY_SHELL_SECTOR || bd.problem < 0 || bd.problem > MGX_CUBE_PROBLEM_SHELL)
    return mgx::report_error(MGX_ERR_INVALID_ARGUMENT, "mgx_cube_create: unknown geometry or problem");
  omp_set_num_threads(effective_threads());
  auto C    = std::make_unique<mgx_cube_s>();
  C->p      = degree;
  C->origin = bd.origin;
  C->h0     = bd.h0;
  C->geometry   = bd.geometry;
  C->problem.id = bd.problem;
  C->rank   = bd.rank;
  C->size   = size;
  int r     = bd.rank;
  for (int d = 0; d < 3; ++d)
    {
      C->groots[d] = bd.roots[d];
      C->procs[d]  = bd.procs[d];
      C->pcoord[d] = r % bd.procs[d];
      r /= bd.procs[d];
      C->lroots[d] = bd.roots[d] / bd.procs[d];
      C->roff[d]   = C->pcoord[d] * C->lroots[d];
    }
  C->brick_numbering = bd.numbering == MGX_CUBE_NUMBERING_BRICK;
  make_basis(C->basis, degree);
  C->levels.resize(n_refine + 1);
  for (int l = 0; l <= n_refine; ++l)
    {
      build_level(*C, C->levels[l], l);
      build_dof_grid(*C, C->levels[l]);
      build_interfaces(*C, C->levels[l]);
      if (C->mapped())
        build_geometry(*C, C->levels[l]);
      build_bc(*C, C->levels[l]);
    }
  *out = C.release();
  return MGX_OK;
}

int mgx_cube_create(int degree, int n_subdiv, int n_refine, mgx_cube_t *out)
{
  return mgx_cube_create_numbered(degree, n_subdiv, n_refine, MGX_CUBE_NUMBERING_BRICK, out);
}

int mgx_cube_create_numbered(int degree, int n_subdiv, int n_refine, int numbering, mgx_cube_t *out)
{
  // "square" mesh: subdivided_hyper_cube(n_subdiv, -0.9, 1.0) (poisson_cube/program.cc:542)
  if (n_subdiv < 1)
    return MGX_ERR_INVALID_ARGUMENT;
  mgx_cube_box_desc bd{};
  bd.degree   = degree;
  bd.n_refine = n_refine;
  bd.origin   = -0.9;
  bd.h0       = 1.9 / n_subdiv;
  bd.rank     = 0;
  bd.numbering = numbering;
  for (int d = 0; d < 3; ++d)
    {
      bd.roots[d] = n_subdiv;
      bd.procs[d] = 1;
    }
  return create_impl(bd, out);
}

int mgx_cube_create_box(const mgx_cube_box_desc *bd, mgx_cube_t *out)
{
  if (!bd)
    return MGX_ERR_INVALID_ARGUMENT;
  return create_impl(*bd, out);
}

int mgx_cube_create_shell(int degree, int n_coarse, int n_refine, int problem, mgx_cube_t *out)
{
  return mgx_cube_create_shell_ranks(degree, n_coarse, n_refine, problem, 1, 0, out);
}

int mgx_cube_create_shell_ranks(int degree, int n_coarse, int n_refine, int problem, int n_ranks, int rank, mgx_cube_t *out)
{
  if (!out || degree < 1 || degree > MGX_MAX_DEGREE || n_refine < 0 || n_refine > 8)
    return mgx::report_error(MGX_ERR_INVALID_ARGUMENT, "mgx_cube_create_shell: degree must be in 1..9 and n_refine in 0..8");
  if (n_coarse != 6 && n_coarse != 12)
    return mgx::report_error(MGX_ERR_INVALID_ARGUMENT, "mgx_cube_create_shell: 6 or 12 coarse cells (GridGenerator::hyper_shell)");
  if (problem < 0 || problem > MGX_CUBE_PROBLEM_SHELL)
    return mgx::report_error(MGX_ERR_INVALID_ARGUMENT, "mgx_cube_create_shell: unknown problem");
  if ((uint64_t)n_coarse << (3 * n_refine) >= 0x10000000ull)
    return mgx::report_error(MGX_ERR_UNSUPPORTED, "mgx_cube_create_shell: too many cells");
  // What is dealt out: the coarse cells in contiguous shares where n_ranks divides them (rank r holds cells
  // [r n_coarse / n_ranks, (r + 1) n_coarse / n_ranks)); otherwise the cells of level 1 -- 48 or 96 of them, equal shares
  // for 8 (and 16, 24, 32) ranks -- where the mesh is refined at least once: the reference partitions the refined mesh
  // cell by cell as well (poisson_shell/program.cc:249,274).  Level i of the object is then level i + 1 of the whole
  // mesh (mgx_cube_level_offset); the coarse cells exist on the undecomposed copy of the coarse levels only
  // (mgx_solver_set_agglomeration).  Neither: uneven shares of coarse cells (12 cells on 8 ranks: 1, 2, 1, 2, ...).
  int granularity = 0;
  if (n_ranks > 1 && n_coarse % n_ranks != 0 && n_refine >= 1 && (8 * n_coarse) % n_ranks == 0 && n_ranks <= 32)
    granularity = 1;
  if (n_ranks < 1 || n_ranks > (n_coarse << (3 * granularity)) || rank < 0 || rank >= n_ranks)
    return mgx::report_error(MGX_ERR_INVALID_ARGUMENT, "mgx_cube_create_shell: at most one rank per coarse cell (or per cell of "
                                                       "level 1 where those can be dealt out evenly)");
  omp_set_num_threads(effective_threads());
  auto C        = std::make_unique<mgx_cube_s>();
  C->p          = degree;
  C->origin     = 0.;
  C->h0         = 0.5;
  C->geometry   = MGX_CUBE_GEOMETRY_HYPER_SHELL;
  C->problem.id = problem;
  C->groots[0] = C->lroots[0] = n_coarse;
  C->shell      = shell_blocks(n_coarse);
  C->rank       = rank;
  C->size       = n_ranks;
  C->level_offset = granularity;
  make_basis(C->basis, degree);
  C->levels.resize(n_refine + 1 - granularity);
  for (int l = granularity; l <= n_refine; ++l)
    {
      std::string why;
      Level      &mine = C->levels[l - granularity];
      if (n_ranks == 1)
        {
          if (!build_shell_level(*C, mine, l, why))
            return mgx::report_error(MGX_ERR_UNSUPPORTED, ("mgx_cube_create_shell: " + why).c_str());
        }
      else
        {
          // the tables of the whole mesh (every rank builds them; they are small next to the per-point
          // coefficients), then this rank's share of it
          Level      whole;
          const bool ok = build_shell_level(*C, whole, l, why);
          if (!ok)
            return mgx::report_error(MGX_ERR_UNSUPPORTED, ("mgx_cube_create_shell: " + why).c_str());
          localise_shell_level(*C, whole, mine, granularity);
        }
      build_geometry(*C, mine);
      build_bc(*C, mine);
    }
  *out = C.release();
  return MGX_OK;
}

int mgx_cube_level_offset(mgx_cube_t c)
{
  return c ? c->level_offset : 0;
}

int mgx_cube_cell_nodes(mgx_cube_t c, int l, double *out)
{
  if (!c || !out || l < 0 || l >= (int)c->levels.size())
    return MGX_ERR_INVALID_ARGUMENT;
  if (c->geometry != MGX_CUBE_GEOMETRY_HYPER_SHELL)
    return mgx::report_error(MGX_ERR_UNSUPPORTED, "mgx_cube_cell_nodes: multi-block meshes only");
  const Level &L  = c->levels[l];
  const size_t n3 = (size_t)(c->p + 1) * (c->p + 1) * (c->p + 1);
#pragma omp parallel for schedule(static)
  for (uint32_t cell = 0; cell < L.n_cells; ++cell)
    shell_cell_nodes(*c, L, cell, out + 3 * n3 * cell);
  return MGX_OK;
}

const uint8_t *mgx_cube_entity_multiplicity(mgx_cube_t c, int l)
{
  return (c && l >= 0 && l < (int)c->levels.size() && !c->levels[l].ent_mult.empty()) ? c->levels[l].ent_mult.data() : nullptr;
}

int mgx_cube_destroy(mgx_cube_t cube)
{
  delete cube;
  return MGX_OK;
}

int      mgx_cube_n_levels(mgx_cube_t c) { return (int)c->levels.size(); }
int      mgx_cube_degree(mgx_cube_t c) { return c->p; }
uint32_t mgx_cube_n_cells(mgx_cube_t c, int l) { return c->levels[l].n_cells; }
uint32_t mgx_cube_n_dofs(mgx_cube_t c, int l) { return c->levels[l].n_dofs; }
uint32_t mgx_cube_n_constrained(mgx_cube_t c, int l) { return (uint32_t)c->levels[l].constrained.size(); }
uint32_t mgx_cube_cells_per_dim(mgx_cube_t c, int l) { return c->levels[l].N; }
double   mgx_cube_cell_size(mgx_cube_t c, int l) { return c->levels[l].h; }

const uint32_t *mgx_cube_idx27(mgx_cube_t c, int l) { return c->levels[l].idx27.data(); }
const uint32_t *mgx_cube_idx27_plain(mgx_cube_t c, int l) { return c->levels[l].idx27_plain.data(); }
const uint32_t *mgx_cube_constrained(mgx_cube_t c, int l) { return c->levels[l].constrained.data(); }
const uint32_t *mgx_cube_children(mgx_cube_t c, int l) { return l > 0 ? c->levels[l].children.data() : nullptr; }
const uint32_t *mgx_cube_cell_coords(mgx_cube_t c, int l) { return c->levels[l].coords.data(); }
const uint32_t *mgx_cube_dof_grid(mgx_cube_t c, int l) { return c->levels[l].dof_grid.data(); }
int             mgx_cube_rank(mgx_cube_t c) { return c->rank; }
int             mgx_cube_size(mgx_cube_t c) { return c->size; }
int             mgx_cube_n_neighbors(mgx_cube_t c, int l) { return (int)c->levels[l].neighbors.size(); }
int             mgx_cube_neighbor_rank(mgx_cube_t c, int l, int k) { return c->levels[l].neighbors[k].rank; }
uint32_t        mgx_cube_neighbor_count(mgx_cube_t c, int l, int k) { return (uint32_t)c->levels[l].neighbors[k].index.size(); }
const uint32_t *mgx_cube_neighbor_index(mgx_cube_t c, int l, int k) { return c->levels[l].neighbors[k].index.data(); }
uint32_t        mgx_cube_n_shared(mgx_cube_t c, int l) { return (uint32_t)c->levels[l].shared.size(); }
const uint32_t *mgx_cube_shared(mgx_cube_t c, int l) { return c->levels[l].shared.data(); }
uint32_t        mgx_cube_n_not_owned(mgx_cube_t c, int l) { return (uint32_t)c->levels[l].not_owned.size(); }
const uint32_t *mgx_cube_not_owned(mgx_cube_t c, int l) { return c->levels[l].not_owned.data(); }
const uint8_t  *mgx_cube_weight_shift(mgx_cube_t c, int l) { return (l > 0 && !c->levels[l].weight_shift.empty()) ? c->levels[l].weight_shift.data() : nullptr; }
void            mgx_cube_cells_per_dim3(mgx_cube_t c, int l, uint32_t local[3], uint32_t global[3])
{
  for (int d = 0; d < 3; ++d)
    {
      local[d]  = c->levels[l].Nl[d];
      global[d] = c->levels[l].Ng[d];
    }
}
const double *mgx_cube_shape_values(mgx_cube_t c) { return c->basis.S; }
const double *mgx_cube_colloc_grad(mgx_cube_t c) { return c->basis.D; }
const double *mgx_cube_qweights(mgx_cube_t c) { return c->basis.gw; }
const double *mgx_cube_qpoints(mgx_cube_t c) { return c->basis.gq; }
const double *mgx_cube_gll(mgx_cube_t c) { return c->basis.gll; }
const double *mgx_cube_prolong_1d(mgx_cube_t c) { return c->basis.P1; }

const double *mgx_cube_rhs(mgx_cube_t c, int l)
{
  if (c->levels[l].rhs.empty()) // assembled when first asked for
    build_rhs(*c, c->levels[l]);
  return c->levels[l].rhs.data();
}

int mgx_cube_rhs_quadrature(mgx_cube_t c, int l, double *out)
{
  if (!c || !out || l < 0 || l >= (int)c->levels.size())
    return mgx::report_error(MGX_ERR_INVALID_ARGUMENT, "mgx_cube_rhs_quadrature: bad argument");
  const Level &L  = c->levels[l];
  const size_t n3 = (size_t)(c->p + 1) * (c->p + 1) * (c->p + 1);
#pragma omp parallel for schedule(static)
  for (uint32_t cell = 0; cell < L.n_cells; ++cell)
    rhs_quadrature_cell(*c, L, cell, out + (size_t)cell * n3);
  return MGX_OK;
}
uint32_t        mgx_cube_bc_count(mgx_cube_t c, int l) { return (uint32_t)c->levels[l].bc_index.size(); }
const uint32_t *mgx_cube_bc_index(mgx_cube_t c, int l) { return c->levels[l].bc_index.data(); }
const double   *mgx_cube_bc_value(mgx_cube_t c, int l) { return c->levels[l].bc_value.data(); }

const double *mgx_cube_coef_q(mgx_cube_t c, int l)
{
  return (c && l >= 0 && l < (int)c->levels.size() && !c->levels[l].coef_q.empty()) ? c->levels[l].coef_q.data() : nullptr;
}

int mgx_cube_operator_desc(mgx_cube_t c, int l, int number, mgx_operator_desc *d)
{
  if (!c || !d || l < 0 || l >= (int)c->levels.size())
    return MGX_ERR_INVALID_ARGUMENT;
  const Level &L   = c->levels[l];
  d->degree        = c->p;
  d->number        = number;
  d->n_cells       = L.n_cells;
  d->n_dofs        = L.n_dofs;
  d->idx27         = L.idx27.data();
  d->idx27_plain   = L.idx27_plain.data();
  d->constrained   = L.constrained.data();
  d->n_constrained = (uint32_t)L.constrained.size();
  // merged_coefficient = a JxW J^-T J^-1 = h^3/h^2 on the Cartesian mesh (laplace_operator.h:374-387)
  d->coef[0] = d->coef[1] = d->coef[2] = L.h;
  d->coef[3] = d->coef[4] = d->coef[5] = 0.;
  d->coef_q       = L.coef_q.empty() ? nullptr : L.coef_q.data(); // general branch (mapped mesh / variable coefficient)
  d->shape_values = c->basis.S;
  d->colloc_grad  = c->basis.D;
  d->qweights     = c->basis.gw;
  d->brick_colour = L.brick_colour.empty() ? nullptr : L.brick_colour.data();
  d->global_index = L.dof_grid.data();
  d->exchange     = nullptr; // decomposed meshes: see mgx_cube_exchange_desc
  return MGX_OK;
}

int mgx_cube_exchange_desc(mgx_cube_t c, int l, int plan_id, mgx_exchange_desc *e, const uint32_t **index_scratch,
                           int *rank_scratch, uint32_t *count_scratch)
{
  if (!c || !e || l < 0 || l >= (int)c->levels.size())
    return MGX_ERR_INVALID_ARGUMENT;
  const Level &L = c->levels[l];
  for (size_t k = 0; k < L.neighbors.size(); ++k)
    {
      index_scratch[k] = L.neighbors[k].index.data();
      rank_scratch[k]  = L.neighbors[k].rank;
      count_scratch[k] = (uint32_t)L.neighbors[k].index.size();
    }
  e->plan_id       = plan_id;
  e->n_neighbors   = (int)L.neighbors.size();
  e->neighbor_rank = rank_scratch;
  e->count         = count_scratch;
  e->index         = index_scratch;
  e->shared        = L.shared.data();
  e->n_shared      = (uint32_t)L.shared.size();
  e->not_owned     = L.not_owned.data();
  e->n_not_owned   = (uint32_t)L.not_owned.size();
  e->send_buf      = nullptr;
  e->recv_buf      = nullptr;
  return MGX_OK;
}

double mgx_cube_l2_error(mgx_cube_t c, int l, const double *sol)
{
  double err2 = 0, vol = 0;
  mgx_cube_l2_error_parts(c, l, sol, &err2, &vol);
  return std::sqrt(err2 / vol);
}

void mgx_cube_l2_error_parts(mgx_cube_t c, int l, const double *sol, double *err2_out, double *vol_out)
{
  const Level &L = c->levels[l];
  const Basis &B = c->basis;
  const int    p = c->p, n = p + 1, n3 = n * n * n;
  const double h = L.h, h3 = h * h * h;
  double       err = 0, vol = 0;
#pragma omp parallel reduction(+ : err, vol)
  {
    std::vector<double> buf(2 * (size_t)n3);
    double             *u = buf.data(), *t0 = u + n3;
#pragma omp for schedule(static)
    for (uint32_t cell = 0; cell < L.n_cells; ++cell)
      {
        gather_cell(p, &L.idx27_plain[27 * (size_t)cell], sol, u); // read_dof_values_plain
        apply_1d(n, 0, B.S, false, u, t0, false);
        apply_1d(n, 1, B.S, false, t0, u, false);
        apply_1d(n, 2, B.S, false, u, t0, false);
        const double x0 = c->origin + h * (L.off[0] + L.coords[3 * (size_t)cell]),
                     y0 = c->origin + h * (L.off[1] + L.coords[3 * (size_t)cell + 1]),
                     z0 = c->origin + h * (L.off[2] + L.coords[3 * (size_t)cell + 2]);
        for (int k = 0, q = 0; k < n; ++k)
          for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i, ++q)
              {
                double JxW = B.gw[i] * B.gw[j] * B.gw[k] * h3, d;
                if (!L.jxw.empty())
                  {
                    const double *xp = &L.xq[((size_t)cell * n3 + q) * 3];
                    JxW              = L.jxw[(size_t)cell * n3 + q];
                    d                = t0[q] - c->problem.u(xp[0], xp[1], xp[2]);
                  }
                else
                  d = t0[q] - c->problem.u(x0 + h * B.gq[i], y0 + h * B.gq[j], z0 + h * B.gq[k]);
                err += d * d * JxW;
                vol += JxW;
              }
      }
  }
  *err2_out = err;
  *vol_out  = vol;
}

int mgx_cube_seeded_vector(mgx_cube_t c, int l, uint64_t seed, double *out)
{
  if (!c || !out || l < 0 || l >= (int)c->levels.size())
    return MGX_ERR_INVALID_ARGUMENT;
  const uint32_t *g = mgx_cube_dof_grid(c, l);
  const uint32_t  n = c->levels[l].n_dofs;
#pragma omp parallel for schedule(static)
  for (uint32_t i = 0; i < n; ++i)
    {
      uint64_t z = seed + 0x9E3779B97F4A7C15ull * ((uint64_t)g[i] + 1);
      z          = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z          = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      z ^= z >> 31;
      out[i] = (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
    }
  return MGX_OK;
}

int mgx_cube_solver_destroy(mgx_cube_solver *s)
{
  if (!s)
    return MGX_OK;
  mgx_solver_destroy(s->solver);
  // transfers reference their level operators: release them first
  for (int l = 0; l < s->n_levels; ++l)
    {
      if (s->transfer && s->transfer_dp && s->transfer_dp[l] != s->transfer[l])
        mgx_transfer_destroy(s->transfer_dp[l]);
      if (s->transfer)
        mgx_transfer_destroy(s->transfer[l]);
    }
  for (int l = 0; l < s->n_levels; ++l)
    {
      if (s->matrix && s->matrix_dp && s->matrix_dp[l] != s->matrix[l])
        mgx_operator_destroy(s->matrix_dp[l]);
      if (s->matrix)
        mgx_operator_destroy(s->matrix[l]);
    }
  delete[] s->matrix;
  delete[] s->matrix_dp;
  delete[] s->transfer;
  delete[] s->transfer_dp;
  std::memset(s, 0, sizeof(*s));
  return MGX_OK;
}

int mgx_cube_solver_create(mgx_context_t ctx, mgx_cube_t cube, int vnumber, int degree_pre, int n_cycles, mgx_cube_solver *out)
{
  return mgx_cube_solver_create_opt(ctx, cube, vnumber, degree_pre, n_cycles, 0, out);
}

int mgx_cube_solver_create_opt(mgx_context_t ctx, mgx_cube_t cube, int vnumber, int degree_pre, int n_cycles, int device_rhs,
                           mgx_cube_solver *out)
{
  if (!ctx || !cube || !out || (vnumber != MGX_F32 && vnumber != MGX_F64))
    return MGX_ERR_INVALID_ARGUMENT;
  const int nl = (int)cube->levels.size();
  std::memset(out, 0, sizeof(*out));
  out->n_levels    = nl;
  out->matrix      = new mgx_operator_t[nl]();
  out->matrix_dp   = new mgx_operator_t[nl]();
  out->transfer    = new mgx_transfer_t[nl]();
  out->transfer_dp = new mgx_transfer_t[nl]();
  int status       = MGX_OK;
  for (int l = 0; l < nl && status == MGX_OK; ++l)
    {
      mgx_operator_desc d;
      mgx_cube_operator_desc(cube, l, MGX_F64, &d);
      // decomposed mesh: plan ids 2*level (fp64 operator) and 2*level+1 (V-cycle operator)
      mgx_exchange_desc ex;
      const uint32_t   *idx[27];
      int               ranks[27];
      uint32_t          counts[27];
      if (cube->size > 1)
        {
          mgx_cube_exchange_desc(cube, l, 2 * l, &ex, idx, ranks, counts);
          d.exchange = &ex;
        }
      status = mgx_operator_create(ctx, &d, &out->matrix_dp[l]);
      if (status != MGX_OK)
        break;
      if (vnumber == MGX_F64)
        out->matrix[l] = out->matrix_dp[l];
      else
        {
          d.number   = MGX_F32;
          ex.plan_id = 2 * l + 1;
          status     = mgx_operator_create(ctx, &d, &out->matrix[l]);
        }
    }
  for (int l = 1; l < nl && status == MGX_OK; ++l)
    {
      mgx_transfer_desc t;
      t.children     = cube->levels[l].children.data();
      t.prolong_1d   = cube->basis.P1;
      t.weight_shift = cube->levels[l].weight_shift.empty() ? nullptr : cube->levels[l].weight_shift.data();
      status       = mgx_transfer_create(out->matrix_dp[l - 1], out->matrix_dp[l], &t, &out->transfer_dp[l]);
      if (status != MGX_OK)
        break;
      if (vnumber == MGX_F64)
        out->transfer[l] = out->transfer_dp[l];
      else
        status = mgx_transfer_create(out->matrix[l - 1], out->matrix[l], &t, &out->transfer[l]);
    }
  if (status == MGX_OK)
    {
      std::vector<const double *>   rhs(nl), bcv(nl);
      std::vector<const uint32_t *> bci(nl);
      std::vector<uint32_t>         bcn(nl);
      for (int l = 0; l < nl; ++l)
        {
          rhs[l] = device_rhs ? nullptr : mgx_cube_rhs(cube, l);
          bci[l] = cube->levels[l].bc_index.data();
          bcv[l] = cube->levels[l].bc_value.data();
          bcn[l] = (uint32_t)cube->levels[l].bc_index.size();
        }
      mgx_solver_desc sd;
      sd.n_levels    = nl;
      sd.degree_pre  = degree_pre;
      sd.n_cycles    = n_cycles;
      sd.matrix      = out->matrix;
      sd.matrix_dp   = out->matrix_dp;
      sd.transfer    = out->transfer;
      sd.transfer_dp = out->transfer_dp;
      sd.rhs         = rhs.data();
      sd.bc_index    = bci.data();
      sd.bc_value    = bcv.data();
      sd.bc_count    = bcn.data();
      status         = mgx_solver_create(ctx, &sd, &out->solver);
      // the right-hand sides on the device (laplace_operator.h:804-845): the host only evaluates f JxW at the
      // quadrature points
      for (int l = 0; l < nl && status == MGX_OK && device_rhs; ++l)
        {
          const size_t        n3 = (size_t)(cube->p + 1) * (cube->p + 1) * (cube->p + 1), count = n3 * cube->levels[l].n_cells;
          std::vector<double> fq(count);
          void               *fq_dev = nullptr;
          status                     = mgx_cube_rhs_quadrature(cube, l, fq.data());
          if (status == MGX_OK)
            status = mgx_malloc(ctx, &fq_dev, sizeof(double) * count);
          if (status == MGX_OK)
            status = mgx_upload(ctx, fq_dev, fq.data(), sizeof(double) * count);
          if (status == MGX_OK)
            status = mgx_solver_compute_rhs(out->solver, l, (const double *)fq_dev);
          if (fq_dev)
            {
              const std::string keep = status != MGX_OK ? mgx_last_error() : "";
              (void)mgx_sync(ctx);
              (void)mgx_free(ctx, fq_dev);
              if (status != MGX_OK)
                mgx::report_error(status, keep.c_str());
            }
        }
    }
  if (status != MGX_OK)
    {
      const std::string keep = mgx_last_error(); // the destroy calls below must not hide the cause
      mgx_cube_solver_destroy(out);
      mgx::report_error(status, keep.c_str());
    }
  return status;
}

} // extern "C"
