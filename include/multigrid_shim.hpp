// multigrid_shim.hpp -- header-only C++ mirror of the reference's operator / solver interface on top
// of the C ABI of mgx.h, so that driver code written against
//     multigrid::LaplaceOperator<dim,fe_degree,number>          common/laplace_operator.h:56-164
//     multigrid::MultigridSolver<dim,fe_degree,Number,Number2>  common/multigrid_solver.h:96-782
// keeps its call sites (vmult, vmult_residual, compute_diagonal, solve, solve_cg, do_matvec,
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

#include <array>
#include <cstddef>
#include <cstdio>
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
    void compute_diagonal() { check(mgx_compute_diagonal(h_)); }
    mgx_operator_t handle() const { return h_; }

  private:
    const Context *ctx_   = nullptr;
    mgx_operator_t h_     = nullptr;
    bool           owned_ = false;
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
} // namespace multigrid
