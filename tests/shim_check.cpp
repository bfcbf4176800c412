// shim_check.cpp -- calls every member of include/multigrid_shim.hpp once (the C++ mirror of the reference's
// LaplaceOperator / MultigridSolver / LaplaceOperatorCompactCombine / JacobiTransformed / MultigridSolverDG interface)
// and prints one "name value" line per result; tests/test_gpu_shim.py compiles it against libmgx.so, runs it and compares
// every value with the same call made through ctypes.  Test infrastructure, not part of the library.
#include "multigrid_shim.hpp"

#include <cmath>
#include <cstdio>
#include <vector>

using namespace multigrid;

static void say(const char *name, double v) { std::printf("%s %.17g\n", name, v); }

template <typename Number>
static Vector<Number> seeded(const Context &ctx, const CubeDiscretization &disc, int level, unsigned long long seed)
{
  std::vector<double> h(disc.n_dofs(level));
  check(mgx_cube_seeded_vector(disc.handle(), level, seed, h.data()));
  std::vector<Number> c(h.begin(), h.end());
  Vector<Number>      v(ctx, c.size());
  v.upload(c);
  return v;
}

int main()
{
  try
    {
      Context ctx(0);
      // ---- LaplaceOperator<3,4,double> on 8^3 cells ----
      {
        CubeDiscretization disc(4, 1, 3);
        const int          level = disc.n_levels() - 1;
        LaplaceOperator<3, 4, double> op;
        op.initialize(ctx, disc, level);
        say("op_m", (double)op.m());
        Vector<double> x = seeded<double>(ctx, disc, level, 1), b = seeded<double>(ctx, disc, level, 2), y, r;
        op.initialize_dof_vector(y);
        op.initialize_dof_vector(r);
        op.vmult(y, x);
        say("vmult_l2", y.l2_norm());
        op.vmult_residual(b, x, r);
        say("vmult_residual_l2", r.l2_norm());
        op.compute_diagonal();
        say("diag_inverse_l2", op.get_matrix_diagonal_inverse().l2_norm());
        // vmult_with_cg_update: x += alpha p, p = beta p + q, q = A p
        Vector<double> q = seeded<double>(ctx, disc, level, 3), p = seeded<double>(ctx, disc, level, 4),
                       xx = seeded<double>(ctx, disc, level, 5);
        const std::array<double, 4> sums = op.vmult_with_cg_update(0.3, 0.7, b, q, p, xx);
        say("cg_update_qp", sums[0]);
        say("cg_update_rr", sums[1]);
        say("cg_update_qr", sums[2]);
        say("cg_update_qq", sums[3]);
        say("cg_update_x_l2", xx.l2_norm());
        // compute_residual with the problem's right-hand side, homogeneous boundary values in src
        Vector<double> u0, rhs;
        op.initialize_dof_vector(u0);
        op.initialize_dof_vector(rhs);
        Vector<double> rhs_q = op.rhs_at_quadrature_points();
        op.compute_residual(rhs, u0, rhs_q);
        say("compute_residual_l2", rhs.l2_norm());
        // evaluate_coefficient: twice the Cartesian coefficient through the per-point branch doubles the product
        {
          // (merged coefficient of a Cartesian cell: det J J^-1 J^-T = h on the diagonal, times the quadrature weight)
          const std::size_t   nq = 125, nc = disc.n_active_cells();
          const double        h  = mgx_cube_cell_size(disc.handle(), level);
          const double       *w  = mgx_cube_qweights(disc.handle());
          std::vector<double> coef(nc * 6 * nq, 0.);
          for (std::size_t c = 0; c < nc; ++c)
            for (int k = 0; k < 3; ++k)
              for (std::size_t i = 0; i < nq; ++i)
                coef[(c * 6 + k) * nq + i] = 2. * h * w[i % 5] * w[(i / 5) % 5] * w[i / 25];
          op.evaluate_coefficient(coef);
          Vector<double> y2;
          op.initialize_dof_vector(y2);
          op.vmult(y2, x);
          // (constrained rows, numbered last, are identity rows of vmult in both operators)
          const std::vector<double> h2 = y2.download(), h1 = y.download();
          const std::size_t         n_free = h1.size() - mgx_cube_n_constrained(disc.handle(), level);
          double                    num = 0, den = 0;
          for (std::size_t i = 0; i < n_free; ++i)
            {
              num += (h2[i] - 2. * h1[i]) * (h2[i] - 2. * h1[i]);
              den += h1[i] * h1[i];
            }
          say("evaluate_coefficient_defect", std::sqrt(num / den));
          op.evaluate_coefficient(std::vector<double>());
        }
      }
      // ---- MultigridSolver<3,4,float,double> ----
      {
        CubeDiscretization disc(4, 1, 3);
        MultigridSolver<3, 4, float, double> solver(ctx, disc, 3, 3, 1);
        say("fmg_reduction", solver.solve(false));
        say("fmg_l2_error", solver.compute_l2_error(disc.n_levels() - 1));
        const auto cg = solver.solve_cg();
        say("cg_its", cg.first);
        say("cg_reduction", cg.second);
        say("cg_l2_error", solver.compute_l2_error(disc.n_levels() - 1));
        say("solution_l2", solver.get_solution().l2_norm());
        const int      level = disc.n_levels() - 1;
        Vector<double> src = seeded<double>(ctx, disc, level, 7), dst(ctx, disc.n_dofs());
        solver.vmult(dst, src);
        say("vcycle_l2", dst.l2_norm());
        Vector<double> res = seeded<double>(ctx, disc, level, 8), upd = seeded<double>(ctx, disc, level, 9);
        const std::array<double, 2> dots = solver.vmult_with_residual_update(res, upd, 0.25);
        say("residual_update_zr", dots[0]);
        say("residual_update_zu", dots[1]);
        solver.do_matvec();
        solver.do_matvec_smoother();
        LaplaceOperator<3, 4, double> A;
        solver.get_operator_dp((unsigned int)level, A);
        A.vmult(dst, src);
        say("solver_operator_vmult_l2", dst.l2_norm());
        ctx.marker_start("matvec");
        ctx.marker_stop("matvec");
      }
      // ---- DG: LaplaceOperatorCompactCombine<3,3,float,0>, JacobiTransformed, MultigridSolverDG ----
      {
        CubeDiscretization disc(3, 1, 3);
        LaplaceOperatorCompactCombine<3, 3, double, 0> dg;
        dg.reinit(ctx, disc);
        say("dg_m", (double)dg.m());
        say("dg_penalty", dg.get_penalty(0, 0));
        Vector<double> x, y, b, o;
        for (Vector<double> *v : {&x, &y, &b, &o})
          dg.initialize_dof_vector(*v);
        std::vector<double> hx(dg.m()), hb(dg.m()), ho(dg.m());
        for (std::size_t i = 0; i < hx.size(); ++i)
          {
            hx[i] = std::sin(0.37 * (double)i);
            hb[i] = std::cos(0.11 * (double)i);
            ho[i] = std::sin(0.05 * (double)i + 1.);
          }
        x.upload(hx);
        b.upload(hb);
        o.upload(ho);
        dg.vmult(y, x);
        say("dg_vmult_l2", y.l2_norm());
        dg.vmult_residual(b, x, y);
        say("dg_vmult_residual_l2", y.l2_norm());
        JacobiTransformed<3, 3, double, 0> jacobi(dg);
        jacobi.vmult(y, x);
        say("dg_jacobi_l2", y.l2_norm());
        dg.vmult_with_chebyshev_update(b, 2, 0.6, 0.2, x, o);
        say("dg_chebyshev_l2", x.l2_norm());
        Vector<double> q, p, xx;
        for (Vector<double> *v : {&q, &p, &xx})
          dg.initialize_dof_vector(*v);
        q.upload(hb);
        p.upload(ho);
        xx.upload(hx);
        const std::array<double, 4> sums = dg.vmult_with_cg_update(0.3, 0.7, b, q, p, xx);
        say("dg_cg_update_qp", sums[0]);
        say("dg_cg_update_qq", sums[3]);
        MultigridSolverDG<3, 3, float, double> mgdg(ctx, disc, 3);
        Vector<double>                         rhs, sol;
        mgdg.matrix_dg_dp.initialize_dof_vector(rhs);
        mgdg.matrix_dg_dp.initialize_dof_vector(sol);
        rhs.upload(hb);
        mgdg.vmult(sol, rhs);
        say("dg_vcycle_l2", sol.l2_norm());
        const auto cg = mgdg.solve_cg(rhs, sol, 1e-9);
        say("dg_cg_its", cg.first);
        say("dg_solution_l2", sol.l2_norm());
        say("dg_smoother_degree", mgdg.smoother_info().degree);
      }
      std::printf("done 1\n");
    }
  catch (const std::exception &e)
    {
      std::printf("error %s\n", e.what());
      return 1;
    }
  return 0;
}
