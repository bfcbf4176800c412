// multigrid_shim.hpp -- header-only C++ mirror of the reference's operator / solver interface on top
// of the C ABI of mgx.h, so that driver code written against
//     multigrid::LaplaceOperator<dim,fe_degree,number>          common/laplace_operator.h:56-164
//     multigrid::MultigridSolver<dim,fe_degree,Number,Number2>  common/multigrid_solver.h:96-782
//     multigrid::LaplaceOperatorCompactCombine<dim,fe_degree,Number,type> + JacobiTransformed
//                                                                common/laplace_operator_dg.h:350-2256
//     multigrid::MultigridSolverDG<dim,fe_degree,Number,Number2> common/multigrid_solver_dg.h:55-747
// keeps its call sites (vmult, vmult_residual, vmult_with_cg_update, compute_residual, evaluate_coefficient,
// compute_diagonal, get_matrix_diagonal_inverse, solve, solve_cg, vmult_with_residual_update, do_matvec,
// compute_l2_error, get_solution, print_wall_times ...).  Non-zero C status codes become
// exceptions, as deal.II's AssertThrow would (SURVEY.md 8b "Errors").
//
// What differs, and why: the reference constructs these classes from deal.II objects
// (MatrixFree, DoFHandler, Function); deal.II is not available to this build, so the constructors
// here take the structured-cube discretisation of mgx_cube.h instead.  INTEGRATION.md shows the
// constructor a deal.II based build would add (it only has to fill mgx_operator_desc /
// mgx_solver_desc from LaplaceOperator::get_compressed_dof_indices() and friends).
#pragma once

#include "mgx.h"
#include "mgx_cube.h"
#include "mgx_dg.h"

#include <array>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace multigrid
{
  struct MgxError : std::runtime_error
  {
    int status;
    MgxError(int status, const std::string &what)
      : std::runtime_error("mgx error " + std::to_string(status) + ": " + what)
      , status(status)
    {}
  };

  inline void check(int status)
  {
    if (status != MGX_OK)
      throw MgxError(status, mgx_last_error());
  }

  template <typename Number>
  struct number_id;
  template <>
  struct number_id<float>
  {
    static constexpr int value = MGX_F32;
  };
  template <>
  struct number_id<double>
  {
    static constexpr int value = MGX_F64;
  };

  // device + stream (one per process/rank, like MPI_InitFinalize in the reference's main)
  class Context
  {
  public:
    explicit Context(int device = 0) { check(mgx_context_create(&h_, device)); }
    ~Context() { mgx_context_destroy(h_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    mgx_context_t handle() const { return h_; }
    void          sync() const { check(mgx_sync(h_)); }
    // options of mgx_context_set_option (before the first object is created on the context)
    void set_option(const char *name, double value) { check(mgx_context_set_option(h_, name, value)); }
    // LIKWID_MARKER_START / STOP of the reference's drivers as profiler ranges (no-ops unless the option "roctx" is set)
    void marker_start(const char *name) const { check(mgx_range_push(h_, name)); }
    void marker_stop(const char * /*name*/) const { check(mgx_range_pop(h_)); }
    // device memory in use on the context's GPU (all processes), in MB
    double device_memory_used_mb() const
    {
      size_t free_b = 0, total_b = 0;
      check(mgx_device_memory_info(h_, &free_b, &total_b));
      return (double)(total_b - free_b) / (1024. * 1024.);
    }

  private:
    mgx_context_t h_ = nullptr;
  };

  // structured-cube discretisation (what Triangulation + DoFHandler + MatrixFree provide)
  class CubeDiscretization
  {
  public:
    CubeDiscretization(int degree, int n_subdiv, int n_refine) { check(mgx_cube_create(degree, n_subdiv, n_refine, &h_)); }
    // the "doubling" mesh family (poisson_cube/program.cc:509-529): subdivisions[d] in {1, 2} coarse
    // cubes of size 1.9 from (-1,-1,-1), refined n_refine times
    CubeDiscretization(int degree, const int (&subdivisions)[3], int n_refine)
    {
      mgx_cube_box_desc bd{};
      bd.degree   = degree;
      bd.n_refine = n_refine;
      bd.origin   = -1.;
      bd.h0       = 1.9;
      bd.rank     = 0;
      bd.numbering = MGX_CUBE_NUMBERING_BRICK;
      for (int d = 0; d < 3; ++d)
        {
          bd.roots[d] = subdivisions[d];
          bd.procs[d] = 1;
        }
      check(mgx_cube_create_box(&bd, &h_));
    }
    ~CubeDiscretization() { mgx_cube_destroy(h_); }
    CubeDiscretization(const CubeDiscretization &) = delete;
    CubeDiscretization &operator=(const CubeDiscretization &) = delete;
    mgx_cube_t   handle() const { return h_; }
    int          n_levels() const { return mgx_cube_n_levels(h_); }
    int          degree() const { return mgx_cube_degree(h_); }
    std::size_t  n_dofs(int level) const { return mgx_cube_n_dofs(h_, level); }
    std::size_t  n_dofs() const { return n_dofs(n_levels() - 1); }
    std::size_t  n_active_cells() const { return mgx_cube_n_cells(h_, n_levels() - 1); }

  private:
    mgx_cube_t h_ = nullptr;
  };

  // LinearAlgebra::distributed::Vector<Number> restricted to what the drivers use
  template <typename Number>
  class Vector
  {
  public:
    Vector() = default;
    Vector(const Context &ctx, std::size_t n) { reinit(ctx, n); }
    Vector(const Context &ctx, Number *borrowed, std::size_t n)
      : ctx_(&ctx)
      , p_(borrowed)
      , n_(n)
      , owned_(false)
    {}
    ~Vector() { clear(); }
    Vector(const Vector &) = delete;
    Vector &operator=(const Vector &) = delete;
    Vector(Vector &&o) noexcept { swap(o); }
    Vector &operator=(Vector &&o) noexcept
    {
      swap(o);
      return *this;
    }
    void reinit(const Context &ctx, std::size_t n)
    {
      clear();
      ctx_ = &ctx;
      n_   = n;
      void *p = nullptr;
      check(mgx_malloc(ctx.handle(), &p, n * sizeof(Number)));
      p_     = static_cast<Number *>(p);
      owned_ = true;
      *this  = Number(0);
    }
    void swap(Vector &o)
    {
      std::swap(ctx_, o.ctx_);
      std::swap(p_, o.p_);
      std::swap(n_, o.n_);
      std::swap(owned_, o.owned_);
    }
    Vector &operator=(Number zero)
    {
      (void)zero; // only `= 0` is used by the reference drivers
      check(mgx_memset_zero(ctx_->handle(), p_, n_ * sizeof(Number)));
      return *this;
    }
    std::size_t   size() const { return n_; }
    std::size_t   locally_owned_size() const { return n_; }
    Number       *begin() { return p_; }
    const Number *begin() const { return p_; }
    double        l2_norm() const
    {
      double r = 0;
      check(mgx_l2_norm(ctx_->handle(), number_id<Number>::value, p_, n_, &r));
      return r;
    }
    double operator*(const Vector &o) const
    {
      double r = 0;
      check(mgx_dot(ctx_->handle(), number_id<Number>::value, p_, o.p_, n_, &r));
      return r;
    }
    void sadd(double s, double a, const Vector &v)
    {
      check(mgx_sadd(ctx_->handle(), number_id<Number>::value, p_, s, a, v.p_, n_));
    }
    template <typename Other>
    void copy_locally_owned_data_from(const Vector<Other> &src)
    {
      check(mgx_copy_cast(ctx_->handle(), p_, number_id<Number>::value, src.begin(), number_id<Other>::value, n_));
    }
    void upload(const std::vector<Number> &host) { check(mgx_upload(ctx_->handle(), p_, host.data(), n_ * sizeof(Number))); }
    std::vector<Number> download() const
    {
      std::vector<Number> host(n_);
      check(mgx_download(ctx_->handle(), host.data(), p_, n_ * sizeof(Number)));
      return host;
    }

  private:
    void clear()
    {
      if (owned_ && p_)
        mgx_free(ctx_->handle(), p_);
      p_     = nullptr;
      owned_ = false;
    }
    const Context *ctx_   = nullptr;
    Number        *p_     = nullptr;
    std::size_t    n_     = 0;
    bool           owned_ = false;
  };

  // multigrid::LaplaceOperator (laplace_operator.h:56-164)
  template <int dim, int fe_degree, typename number>
  class LaplaceOperator
  {
    static_assert(dim == 3, "the MI355X path implements dim = 3 (poisson_cube/program.cc:67)");

  public:
    typedef number value_type;
    LaplaceOperator() = default;
    ~LaplaceOperator() { clear(); }
    LaplaceOperator(const LaplaceOperator &) = delete;
    LaplaceOperator &operator=(const LaplaceOperator &) = delete;

    // initialize(matrix_free, constraints, mg_constrained_dofs, level) + evaluate_coefficient(1.)
    void initialize(const Context &ctx, const CubeDiscretization &disc, unsigned int level)
    {
      clear();
      if (disc.degree() != fe_degree)
        throw MgxError(MGX_ERR_INVALID_ARGUMENT, "LaplaceOperator: fe_degree mismatch");
      ctx_ = &ctx;
      mgx_operator_desc d;
      check(mgx_cube_operator_desc(disc.handle(), (int)level, number_id<number>::value, &d));
      check(mgx_operator_create(ctx.handle(), &d, &h_));
      owned_ = true;
      desc_  = d; // (the tables stay owned by the discretisation)
      disc_  = &disc;
      level_ = (int)level;
    }
    void attach(const Context &ctx, mgx_operator_t h)
    {
      clear();
      ctx_   = &ctx;
      h_     = h;
      owned_ = false;
    }
    void clear()
    {
      if (owned_ && h_)
        mgx_operator_destroy(h_);
      h_ = nullptr;
    }
    std::size_t m() const { return mgx_operator_n_dofs(h_); }
    void        initialize_dof_vector(Vector<number> &v) const { v.reinit(*ctx_, m()); }
    void        vmult(Vector<number> &dst, const Vector<number> &src) const { check(mgx_vmult(h_, dst.begin(), src.begin())); }
    void        vmult_residual(const Vector<number> &rhs, const Vector<number> &lhs, Vector<number> &residual) const
    {
      check(mgx_vmult_residual(h_, rhs.begin(), lhs.begin(), residual.begin()));
    }
    // vmult_with_cg_update(alpha, beta, r, q, p, x) (laplace_operator.h:638-719): x += alpha p, p = beta p + q
    // (alpha == 0: p = q), q = A p in one pass of the cell loop; returns {q.p, r.r, q.r, q.q}
    std::array<number, 4> vmult_with_cg_update(const number alpha, const number beta, const Vector<number> &r, Vector<number> &q,
                                               Vector<number> &p, Vector<number> &x) const
    {
      double sums[4];
      check(mgx_vmult_with_cg_update(h_, (double)alpha, (double)beta, r.begin(), q.begin(), p.begin(), x.begin(), nullptr, sums));
      return {{(number)sums[0], (number)sums[1], (number)sums[2], (number)sums[3]}};
    }
    // compute_residual(dst, src, rhs_function) (laplace_operator.h:804-845): dst = (f, phi) - (grad phi, K grad u_bc) with
    // the boundary values in the constrained entries of src.  The reference evaluates a Function at the quadrature
    // points; here the values f(x_q) JxW_q arrive as a device vector [cell][(p+1)^3] (rhs_at_quadrature_points() gives
    // those of the discretisation's problem)
    void compute_residual(Vector<number> &dst, Vector<number> &src, const Vector<number> &rhs_q) const
    {
      check(mgx_compute_residual(h_, dst.begin(), src.begin(), rhs_q.begin()));
    }
    Vector<number> rhs_at_quadrature_points() const
    {
      const std::size_t   n = (std::size_t)mgx_cube_n_cells(disc_->handle(), level_) * (fe_degree + 1) * (fe_degree + 1) * (fe_degree + 1);
      std::vector<double> host(n);
      check(mgx_cube_rhs_quadrature(disc_->handle(), level_, host.data()));
      std::vector<number> cast(host.begin(), host.end());
      Vector<number>      v(*ctx_, n);
      v.upload(cast);
      return v;
    }
    // evaluate_coefficient(coefficient_function) (laplace_operator.h:357-432): the merged coefficient JxW J^-1 a J^-T
    // (quadrature weight folded in, :388-430) per cell and quadrature point, six entries each ([cell][6][(p+1)^3], host; the reference evaluates a Function and
    // the mapping, here the caller or the discretisation supplies the values).  An empty vector restores the constant
    // Cartesian coefficient of the discretisation.
    void evaluate_coefficient(const std::vector<number> &merged_coefficient)
    {
      if (!owned_)
        throw MgxError(MGX_ERR_UNSUPPORTED, "evaluate_coefficient: operator is borrowed from a solver");
      mgx_operator_desc d = desc_;
      if (!merged_coefficient.empty())
        {
          if (merged_coefficient.size() != (std::size_t)d.n_cells * 6 * (fe_degree + 1) * (fe_degree + 1) * (fe_degree + 1))
            throw MgxError(MGX_ERR_INVALID_ARGUMENT, "evaluate_coefficient: expected [cell][6][(p+1)^3] values");
          d.coef_q = merged_coefficient.data();
        }
      mgx_operator_t fresh = nullptr;
      check(mgx_operator_create(ctx_->handle(), &d, &fresh));
      mgx_operator_destroy(h_);
      h_ = fresh;
    }
    void compute_diagonal() { check(mgx_compute_diagonal(h_)); }
    // Base::get_matrix_diagonal_inverse()->get_vector() (filled by compute_diagonal, laplace_operator.h:745-800): borrowed
    Vector<number> get_matrix_diagonal_inverse() const
    {
      const void *dptr = nullptr;
      check(mgx_get_inverse_diagonal(h_, &dptr));
      return Vector<number>(*ctx_, static_cast<number *>(const_cast<void *>(dptr)), m());
    }
    mgx_operator_t handle() const { return h_; }

  private:
    const Context            *ctx_   = nullptr;
    const CubeDiscretization *disc_  = nullptr;
    mgx_operator_desc         desc_{};
    int                       level_ = 0;
    mgx_operator_t            h_     = nullptr;
    bool                      owned_ = false;
  };

  // multigrid::MultigridSolver (multigrid_solver.h:96-782)
  template <int dim, int fe_degree, typename Number, typename Number2>
  class MultigridSolver
  {
    static_assert(dim == 3, "the MI355X path implements dim = 3");
    static_assert(std::is_same<Number2, double>::value, "the outer iteration is fp64 (program.cc:77)");

  public:
    // reference: MultigridSolver(dof_handler, boundary_values, right_hand_side, coefficient,
    //                            degree_pre, degree_post, n_cycles = 1)   multigrid_solver.h:100-106
    // boundary values / rhs / coefficient of poisson_cube are part of the cube discretisation.
    MultigridSolver(const Context &ctx, const CubeDiscretization &disc, const unsigned int degree_pre,
                    const unsigned int degree_post, const unsigned int n_cycles = 1)
      : ctx_(ctx)
      , disc_(disc)
    {
      if (degree_pre != degree_post) // multigrid_solver.h:126-128
        throw MgxError(MGX_ERR_UNSUPPORTED, "Change of pre- and post-smoother degree currently not possible");
      if (disc.degree() != fe_degree)
        throw MgxError(MGX_ERR_INVALID_ARGUMENT, "MultigridSolver: fe_degree mismatch");
      check(mgx_cube_solver_create(ctx.handle(), disc.handle(), number_id<Number>::value, (int)degree_pre,
                                   (int)n_cycles, &s_));
      maxlevel_ = s_.n_levels - 1;
      // the reference's specialisation MultigridSolver<dim,fe_degree,Number,Number> smooths with
      // Chebyshev polynomials of the fourth kind (multigrid_solver.h:951-952), the general
      // template with the first kind (:277-278)
      if (std::is_same<Number, Number2>::value)
        check(mgx_solver_set_polynomial_type(s_.solver, MGX_CHEBYSHEV_FOURTH_KIND));
    }
    ~MultigridSolver() { mgx_cube_solver_destroy(&s_); }
    MultigridSolver(const MultigridSolver &) = delete;
    MultigridSolver &operator=(const MultigridSolver &) = delete;

    // solve(do_analyze): FMG; returns the V-cycle reduction rate (:387-476)
    double solve(const bool do_analyze)
    {
      double              rate = 1.;
      std::vector<double> trace(2 * (maxlevel_ + 1), 0.);
      if (do_analyze)
        {
          // the reference prints error and residual before and after the cycles of every level
          // (:420-473); the errors are evaluated on the host against the analytic solution at the
          // same two points of the solve (mgx_solver_solve_hooked)
          errors_.assign(2 * (maxlevel_ + 1), 0.);
          check(mgx_solver_solve_hooked(s_.solver, 1, &rate, trace.data(), &MultigridSolver::level_hook, this));
          for (int l = 1; l <= maxlevel_; ++l)
            {
              std::printf("error start         level %d: %g\n", l, errors_[2 * l]);
              std::printf("residual norm start level %d: %g\n", l, trace[2 * l]);
              std::printf("residual norm end   level %d: %g\n", l, trace[2 * l + 1]);
              std::printf("error end           level %d: %g\n", l, errors_[2 * l + 1]);
            }
        }
      else
        check(mgx_solver_solve(s_.solver, 0, &rate, nullptr));
      return rate;
    }
    // solve_cg(): (iterations, reduction per iteration) (:483-493)
    std::pair<unsigned int, double> solve_cg()
    {
      unsigned int its = 0;
      double       red = 1.;
      check(mgx_solver_solve_cg(s_.solver, &its, &red));
      return std::make_pair(its, red);
    }
    // preconditioner interface (:498-510)
    void vmult(Vector<Number2> &dst, const Vector<Number2> &src) const { check(mgx_solver_vmult(s_.solver, dst.begin(), src.begin())); }
    // vmult_with_residual_update(residual, update, factor) (:516-619): the V-cycle as preconditioner with the residual
    // update of the PCG step merged into its two precision casts; returns {z.residual, z.(factor update)}
    std::array<Number2, 2> vmult_with_residual_update(Vector<Number2> &residual, Vector<Number2> &update, const Number2 factor) const
    {
      double out[2];
      check(mgx_solver_vmult_with_residual_update(s_.solver, residual.begin(), update.begin(), (double)factor, out));
      return {{(Number2)out[0], (Number2)out[1]}};
    }
    // the operator of a level in the V-cycle number type / in fp64 (matrix[level], matrix_dp[level]); borrowed
    void get_operator(const unsigned int level, LaplaceOperator<dim, fe_degree, Number> &op) const
    {
      mgx_operator_t h = nullptr;
      check(mgx_solver_get_operator(s_.solver, (int)level, 0, &h));
      op.attach(ctx_, h);
    }
    void get_operator_dp(const unsigned int level, LaplaceOperator<dim, fe_degree, Number2> &op) const
    {
      mgx_operator_t h = nullptr;
      check(mgx_solver_get_operator(s_.solver, (int)level, 1, &h));
      op.attach(ctx_, h);
    }
    void do_matvec() { check(mgx_solver_do_matvec(s_.solver)); }                   // :624-628
    void do_matvec_smoother() { check(mgx_solver_do_matvec_smoother(s_.solver)); } // :633-637
    // L2 errors {start, end} per level of the last solve(true) (:420-424, 468-472)
    const std::vector<double> &level_errors() const { return errors_; }
    // compute_l2_error(level) (:298-343): host evaluation against the analytic solution
    double compute_l2_error(const unsigned int level)
    {
      const double *dptr = nullptr;
      check(mgx_solver_get_solution(s_.solver, (int)level, 1, &dptr));
      std::vector<double> host(disc_.n_dofs((int)level));
      check(mgx_download(ctx_.handle(), host.data(), dptr, host.size() * sizeof(double)));
      return mgx_cube_l2_error(disc_.handle(), (int)level, host.data());
    }
    // get_solution() (:376-382): device vector with the boundary values inserted (borrowed)
    Vector<Number2> get_solution()
    {
      const double *dptr = nullptr;
      check(mgx_solver_get_solution(s_.solver, maxlevel_, 1, &dptr));
      return Vector<Number2>(ctx_, const_cast<double *>(dptr), disc_.n_dofs(maxlevel_));
    }
    // print_wall_times() (:348-371)
    void print_wall_times()
    {
      std::vector<double> t(6 * (maxlevel_ + 1));
      check(mgx_solver_get_timings(s_.solver, t.data()));
      std::printf("Coarse solver %d times: %g tot prec %g\n", (int)t[1], t[0], t[2]);
      std::printf("level  smoother    mg_mv     mg_vec    restrict  prolongate  inhomBC\n");
      for (int l = 1; l <= maxlevel_; ++l)
        std::printf("L%-2d    %-12.4g%-10.4g%-10.4g%-10.4g%-12.4g%-10.4g\n", l, t[6 * l + 5], t[6 * l + 0],
                    t[6 * l + 4], t[6 * l + 1], t[6 * l + 2], t[6 * l + 3]);
    }
    void          enable_timings(bool on) { check(mgx_solver_enable_timings(s_.solver, on ? 1 : 0)); }
    mgx_solver_t  handle() const { return s_.solver; }

  private:
    static void level_hook(void *user, int level, int stage)
    {
      auto *self                          = static_cast<MultigridSolver *>(user);
      self->errors_[2 * level + stage] = self->compute_l2_error((unsigned int)level);
    }
    std::vector<double>       errors_;
    const Context            &ctx_;
    const CubeDiscretization &disc_;
    mgx_cube_solver           s_{};
    int                       maxlevel_ = 0;
  };
  // multigrid::LaplaceOperatorCompactCombine<dim,fe_degree,Number,type> (laplace_operator_dg.h:350-2025) on an affine mesh:
  // type 0 FE_DGQHermite, 1 FE_DGQ on Gauss-Lobatto points, 2 FE_DGQ on Gauss points.  The reference's reinit takes a
  // MatrixFree; here the mesh arrives as the neighbour table of its cells and the one cell Jacobian (the reference asserts
  // a single Jacobian as well, :749-750).
  template <int dim, int fe_degree, typename Number, int type = 0>
  class LaplaceOperatorCompactCombine
  {
    static_assert(dim == 3, "the MI355X path implements dim = 3");
    static_assert(type >= 0 && type <= 2, "Only types=0,1,2 implemented");

  public:
    typedef Number value_type;
    LaplaceOperatorCompactCombine() = default;
    ~LaplaceOperatorCompactCombine() { clear(); }
    LaplaceOperatorCompactCombine(const LaplaceOperatorCompactCombine &) = delete;
    LaplaceOperatorCompactCombine &operator=(const LaplaceOperatorCompactCombine &) = delete;
    // neighbours: [n_cells][6] cell behind face 2d + s or MGX_DG_BOUNDARY; jacobian: dx/dxi, row-major
    void reinit(const Context &ctx, const std::vector<std::int32_t> &neighbours, const double (&jacobian)[9])
    {
      clear();
      ctx_ = &ctx;
      mgx_dg_operator_desc d{};
      d.degree     = fe_degree;
      d.basis      = type;
      d.number     = number_id<Number>::value;
      d.n_cells    = (std::uint32_t)(neighbours.size() / 6);
      d.neighbours = neighbours.data();
      for (int i = 0; i < 9; ++i)
        d.jacobian[i] = jacobian[i];
      check(mgx_dg_operator_create(ctx.handle(), &d, &h_));
    }
    // the Cartesian mesh of a cube discretisation: the cells of its finest level in the provider's order, all outer
    // faces Dirichlet
    void reinit(const Context &ctx, const CubeDiscretization &disc)
    {
      const int            l  = disc.n_levels() - 1;
      const std::uint32_t  nc = mgx_cube_n_cells(disc.handle(), l), N = mgx_cube_cells_per_dim(disc.handle(), l);
      const std::uint32_t *xyz = mgx_cube_cell_coords(disc.handle(), l);
      std::vector<std::int32_t> at((std::size_t)N * N * N, MGX_DG_BOUNDARY), nb((std::size_t)nc * 6);
      for (std::uint32_t c = 0; c < nc; ++c)
        at[((std::size_t)xyz[3 * c + 2] * N + xyz[3 * c + 1]) * N + xyz[3 * c]] = (std::int32_t)c;
      for (std::uint32_t c = 0; c < nc; ++c)
        for (int d = 0; d < 3; ++d)
          for (int s = 0; s < 2; ++s)
            {
              std::int64_t p[3] = {xyz[3 * c], xyz[3 * c + 1], xyz[3 * c + 2]};
              p[d] += s ? 1 : -1;
              nb[6 * (std::size_t)c + 2 * d + s] =
                (p[d] < 0 || p[d] >= (std::int64_t)N) ? MGX_DG_BOUNDARY : at[((std::size_t)p[2] * N + p[1]) * N + p[0]];
            }
      const double h      = mgx_cube_cell_size(disc.handle(), l);
      const double jac[9] = {h, 0, 0, 0, h, 0, 0, 0, h};
      reinit(ctx, nb, jac);
    }
    void clear()
    {
      if (h_)
        mgx_dg_operator_destroy(h_);
      h_ = nullptr;
    }
    std::size_t m() const { return (std::size_t)mgx_dg_operator_n_dofs(h_); }
    void        initialize_dof_vector(Vector<Number> &v) const { v.reinit(*ctx_, (std::size_t)mgx_dg_operator_vector_size(h_)); }
    double      get_penalty(const unsigned int /*cell*/, const unsigned int face) const
    {
      double pen[3];
      check(mgx_dg_operator_info(h_, nullptr, pen, nullptr));
      return pen[face / 2];
    }
    void vmult(Vector<Number> &dst, const Vector<Number> &src) const { check(mgx_dg_vmult(h_, dst.begin(), src.begin())); }
    void vmult_residual(const Vector<Number> &rhs, const Vector<Number> &lhs, Vector<Number> &residual) const
    {
      check(mgx_dg_vmult_residual(h_, residual.begin(), rhs.begin(), lhs.begin()));
    }
    // :863-908
    std::array<Number, 4> vmult_with_cg_update(const Number alpha, const Number beta, const Vector<Number> &r, Vector<Number> &q,
                                               Vector<Number> &p, Vector<Number> &x) const
    {
      double sums[4];
      check(mgx_dg_vmult_with_cg_update(h_, (double)alpha, (double)beta, r.begin(), q.begin(), p.begin(), x.begin(), sums));
      return {{(Number)sums[0], (Number)sums[1], (Number)sums[2], (Number)sums[3]}};
    }
    // :910-955; the two vectors trade their storage for iteration_index >= 1, as the reference's swap does (:931)
    void vmult_with_chebyshev_update(const Vector<Number> &rhs, const unsigned int iteration_index, const Number factor1,
                                     const Number factor2, Vector<Number> &solution, Vector<Number> &solution_old) const
    {
      check(mgx_dg_vmult_with_chebyshev_update(h_, rhs.begin(), iteration_index, (double)factor1, (double)factor2,
                                               solution.begin(), solution_old.begin()));
      if (iteration_index > 0)
        solution.swap(solution_old);
    }
    mgx_dg_operator_t handle() const { return h_; }
    const Context    &context() const { return *ctx_; }

  private:
    const Context    *ctx_ = nullptr;
    mgx_dg_operator_t h_   = nullptr;
  };

  // multigrid::JacobiTransformed (laplace_operator_dg.h:2028-2256): block Jacobi in the eigenvector basis of the cell
  template <int dim, int fe_degree, typename Number, int type = 0>
  class JacobiTransformed
  {
  public:
    explicit JacobiTransformed(const LaplaceOperatorCompactCombine<dim, fe_degree, Number, type> &laplace)
      : laplace_(laplace)
    {}
    std::size_t m() const { return laplace_.m(); }
    void        vmult(Vector<Number> &dst, const Vector<Number> &src) const
    {
      check(mgx_dg_jacobi_vmult(laplace_.handle(), dst.begin(), src.begin()));
    }

  private:
    const LaplaceOperatorCompactCombine<dim, fe_degree, Number, type> &laplace_;
  };

  // multigrid::MultigridSolverDG<dim,fe_degree,Number,Number2> (multigrid_solver_dg.h:55-747): the DG level on top of the
  // FE_Q(fe_degree) hierarchy of the same mesh
  template <int dim, int fe_degree, typename Number, typename Number2, int type = 0>
  class MultigridSolverDG
  {
    static_assert(std::is_same<Number2, double>::value, "the outer iteration is fp64");

  public:
    MultigridSolverDG(const Context &ctx, const CubeDiscretization &disc, const unsigned int degree_pre)
      : cfe_(ctx, disc, degree_pre, degree_pre, 1)
    {
      matrix_dg.reinit(ctx, disc);
      matrix_dg_dp.reinit(ctx, disc);
      mgx_dg_solver_desc d{};
      d.matrix_dg    = matrix_dg.handle();
      d.matrix_dg_dp = matrix_dg_dp.handle();
      d.cfe          = cfe_.handle();
      d.degree_pre   = (int)degree_pre;
      // decomposition-independent cell ids (lexicographic position in the mesh): the start vector of the smoother's
      // eigenvalue estimate is tied to them, as deal.II ties it to the global DoF index
      const int                  l   = disc.n_levels() - 1;
      const std::uint32_t        nc  = mgx_cube_n_cells(disc.handle(), l), N = mgx_cube_cells_per_dim(disc.handle(), l);
      const std::uint32_t       *xyz = mgx_cube_cell_coords(disc.handle(), l);
      std::vector<std::uint32_t> gid(nc);
      for (std::uint32_t c = 0; c < nc; ++c)
        gid[c] = xyz[3 * c] + N * (xyz[3 * c + 1] + N * xyz[3 * c + 2]);
      d.cell_global_id = gid.data();
      check(mgx_dg_solver_create(ctx.handle(), &d, &h_));
    }
    ~MultigridSolverDG() { mgx_dg_solver_destroy(h_); }
    MultigridSolverDG(const MultigridSolverDG &) = delete;
    MultigridSolverDG &operator=(const MultigridSolverDG &) = delete;
    // vmult (:429-440): one DG V-cycle
    void vmult(Vector<Number2> &dst, const Vector<Number2> &src) const { check(mgx_dg_solver_vmult(h_, dst.begin(), src.begin())); }
    // solve_cg(tolerance) (:410-424) on a given right-hand side: (iterations, reduction per iteration)
    std::pair<unsigned int, double> solve_cg(const Vector<Number2> &rhs, Vector<Number2> &solution, const double tolerance = 1e-9)
    {
      unsigned int its = 0;
      double       red = 1.;
      check(mgx_dg_solver_solve_cg(h_, tolerance, rhs.begin(), solution.begin(), &its, &red));
      return std::make_pair(its, red);
    }
    mgx_smoother_info smoother_info() const
    {
      mgx_smoother_info i{};
      check(mgx_dg_solver_smoother_info(h_, &i));
      return i;
    }
    LaplaceOperatorCompactCombine<dim, fe_degree, Number, type>  matrix_dg;    // :700
    LaplaceOperatorCompactCombine<dim, fe_degree, Number2, type> matrix_dg_dp; // :703
    mgx_dg_solver_t handle() const { return h_; }

  private:
    MultigridSolver<dim, fe_degree, Number, Number2> cfe_; // FE_Q hierarchy, re-configured by the DG solver (:271-291)
    mgx_dg_solver_t                                  h_ = nullptr;
  };
} // namespace multigrid
