// poisson_cube.cpp -- the poisson_cube benchmark driver on the MI355X path.
//
// Mirrors poisson_cube/program.cc of the reference: same command line
//     ./poisson_cube degree minsize maxsize n_mg_cycles n_pre_smooth n_post_smooth [d|s]
// (program.cc:666-699), same mesh sequence (:498-500, :532-545), same measurement protocol
// (7x FMG best-of, 1x analysed solve, 10x PCG best-of, 5 x (200|50) matvecs, :285-380) and the
// same output lines and final table (:360-362, :382-388, :580-606) -- written against the
// classes of multigrid_shim.hpp instead of deal.II.
//
// Differences: the V-cycle number type is a run-time choice (8th argument: f32 = reference default,
// f64); the per-level wall times (print_wall_times) cover the analysed solve only -- the level
// timers synchronise the stream, which the seven timed solves must not pay for (the reference's
// CPU timers run through all eight).  The analysed solve prints the reference's four lines per level (error start / residual
// start / residual end / error end, multigrid_solver.h:420-473).
#include "../../include/multigrid_shim.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

namespace
{
  using clock_type = std::chrono::steady_clock;
  double seconds_since(clock_type::time_point t0) { return std::chrono::duration<double>(clock_type::now() - t0).count(); }

  struct Row
  {
    std::size_t  cells, dofs;
    double       mv_outer, mv_inner, reduction, fmg_error, fmg_time, cg_error, cg_time, cg_reduction;
    unsigned int cg_its;
  };

  template <int degree, typename vcycle_number>
  void solve(const multigrid::Context &ctx, const multigrid::CubeDiscretization &disc, unsigned int n_mg_cycles,
             unsigned int n_pre, unsigned int n_post, std::vector<Row> &table)
  {
    using namespace multigrid;
    auto t_setup = clock_type::now();
    MultigridSolver<3, degree, vcycle_number, double> solver(ctx, disc, n_pre, n_post, n_mg_cycles);
    ctx.sync();
    std::cout << "Time setup solver (rhs, smoother, device upload): " << seconds_since(t_setup) << std::endl;
    {
      // program.cc:273-279 (one process: min = avg = max); the device footprint next to it
      double rss_mb = 0;
      if (FILE *f = std::fopen("/proc/self/status", "r"))
        {
          char line[256];
          while (std::fgets(line, sizeof(line), f))
            if (std::strncmp(line, "VmRSS:", 6) == 0)
              rss_mb = std::atof(line + 6) / 1024.;
          std::fclose(f);
        }
      std::cout << "Memory stats [MB]: " << rss_mb << " [p0] " << rss_mb << " " << rss_mb << " [p0]   device: "
                << ctx.device_memory_used_mb() << std::endl;
    }
    double best_time = 1e10, tot_time = 0;
    ctx.marker_start("fmg_solver"); // program.cc:282
    for (unsigned int i = 0; i < 7; ++i) // program.cc:285-293
      {
        auto t0 = clock_type::now();
        solver.solve(false);
        ctx.sync();
        const double t = seconds_since(t0);
        best_time      = std::min(t, best_time);
        tot_time += t;
        std::cout << "Time solve                 " << t << "\n";
      }
    ctx.marker_stop("fmg_solver"); // :295
    solver.enable_timings(true);
    const double vcycl_reduction = solver.solve(true); // :297
    std::cout << "All solver time " << tot_time << std::endl;
    solver.print_wall_times(); // :302
    solver.enable_timings(false);
    const unsigned int maxlevel = disc.n_levels() - 1;
    const double       l2_error = solver.compute_l2_error(maxlevel); // :304
    std::cout << "Solution l2 norm = " << solver.get_solution().l2_norm() << " error = " << l2_error << std::endl;
    double                          time_cg = 1e10;
    std::pair<unsigned int, double> cg_details;
    ctx.marker_start("cg_solver"); // :309
    for (unsigned int i = 0; i < 10; ++i) // :313-319
      {
        auto t0    = clock_type::now();
        cg_details = solver.solve_cg();
        ctx.sync();
        const double t = seconds_since(t0);
        time_cg        = std::min(t, time_cg);
        std::cout << "Time solve CG              " << t << "\n";
      }
    ctx.marker_stop("cg_solver"); // :321
    const double l2_error_cg = solver.compute_l2_error(maxlevel); // :323
    const std::size_t n_dofs = disc.n_dofs();
    double            best_mv = 1e10;
    for (unsigned int i = 0; i < 5; ++i) // :343-363
      {
        const unsigned int n_mv = n_dofs < 10000000 ? 200 : 50;
        ctx.sync();
        auto t0 = clock_type::now();
        ctx.marker_start("matvec"); // :348
        for (unsigned int j = 0; j < n_mv; ++j)
          solver.do_matvec();
        ctx.marker_stop("matvec"); // :354
        ctx.sync();
        const double t = seconds_since(t0) / n_mv;
        best_mv        = std::min(best_mv, t);
        std::cout << "matvec time dp " << t << " [p0] " << t << " " << t << " [p0] DoFs/s: " << n_dofs / t << std::endl;
      }
    double best_mvs = 1e10;
    for (unsigned int i = 0; i < 5; ++i) // :364-380
      {
        const unsigned int n_mv = n_dofs < 10000000 ? 200 : 50;
        ctx.sync();
        auto t0 = clock_type::now();
        ctx.marker_start("matvec_sp"); // :369
        for (unsigned int j = 0; j < n_mv; ++j)
          solver.do_matvec_smoother();
        ctx.marker_stop("matvec_sp"); // :375
        ctx.sync();
        best_mvs = std::min(best_mvs, seconds_since(t0) / n_mv);
      }
    std::cout << "Best timings for ndof = " << n_dofs << "   mv " << best_mv << "    mv smooth " << best_mvs
              << "   fmg " << best_time << "   cg-mg " << time_cg << std::endl;
    std::cout << "L2 error with ndof = " << n_dofs << "  " << l2_error << "  with CG " << l2_error_cg << std::endl;
    table.push_back({disc.n_active_cells(), n_dofs, best_mv, best_mvs, vcycl_reduction, l2_error, best_time,
                     l2_error_cg, time_cg, cg_details.second, cg_details.first});
  }

  template <int degree, typename vcycle_number>
  void run(std::size_t min_size, std::size_t max_size, unsigned int n_mg_cycles, unsigned int n_pre,
           unsigned int n_post, bool use_doubling_mesh)
  {
    std::cout << "Testing FE_Q<3>(" << degree << ")" << std::endl;
    multigrid::Context ctx(0);
    if (std::getenv("MGX_ROCTX")) // profiler ranges in place of the reference's LIKWID build (-DLIKWID_PERFMON)
      ctx.set_option("roctx", 1.);
    const unsigned int sizes[] = {1,   2,   3,   4,   5,   6,   7,   8,   10,  12,  14,   16,   20,
                                  24,  28,  32,  40,  48,  56,  64,  80,  96,  112, 128,  160,  192,
                                  224, 256, 320, 384, 448, 512, 640, 768, 896, 1024, 1280, 1536}; // :498-500
    std::vector<Row>   table;
    for (unsigned int cycle = 0; cycle < sizeof(sizes) / sizeof(unsigned int); ++cycle)
      {
        std::cout << "Cycle " << cycle << std::endl;
        unsigned int n_refine = 0, n_subdiv = sizes[cycle]; // :532-539
        int          subdivisions[3] = {1, 1, 1};
        std::size_t  projected_size  = 1;
        if (use_doubling_mesh) // :509-529
          {
            n_refine = cycle / 3;
            for (unsigned int d = 0; d < cycle % 3; ++d)
              subdivisions[d] = 2;
            for (unsigned int d = 0; d < 3; ++d)
              projected_size *= (std::size_t)(1u << n_refine) * subdivisions[d] * degree + 1;
          }
        else
          {
            if (n_subdiv > 1)
              while (n_subdiv % 2 == 0)
                {
                  n_refine += 1;
                  n_subdiv /= 2;
                }
            const std::size_t n1 = (std::size_t)(1u << n_refine) * n_subdiv * degree + 1;
            projected_size       = n1 * n1 * n1; // :544-545
          }
        if (projected_size < min_size)
          continue;
        if (projected_size > max_size)
          {
            std::cout << "Projected size " << projected_size << " higher than max size, terminating." << std::endl
                      << std::endl;
            break;
          }
        auto t0 = clock_type::now();
        if (n_refine > 9)
          {
            std::cout << "More than 9 refinements are not supported, terminating." << std::endl << std::endl;
            break;
          }
        std::unique_ptr<multigrid::CubeDiscretization> disc_ptr;
        if (use_doubling_mesh)
          disc_ptr = std::make_unique<multigrid::CubeDiscretization>(degree, subdivisions, (int)n_refine);
        else
          disc_ptr = std::make_unique<multigrid::CubeDiscretization>(degree, (int)n_subdiv, (int)n_refine);
        const multigrid::CubeDiscretization &disc = *disc_ptr;
        std::cout << "Number of degrees of freedom: " << disc.n_dofs() << std::endl; // :218-220
        std::cout << "DoF setup time:        " << seconds_since(t0) << "s" << std::endl;
        solve<degree, vcycle_number>(ctx, disc, n_mg_cycles, n_pre, n_post, table);
        std::cout << std::endl;
      }
    // ConvergenceTable of program.cc:580-606 (rates as reduction_rate_log2 in dim = 3)
    std::printf("%-9s %-11s %-10s %-10s %-10s %-10s %-5s %-10s %-10s %-5s %-10s %-6s %-12s\n", "cells", "dofs", "mv_outer",
                "mv_inner", "reduction", "fmg_L2error", "", "fmg_time", "cg_L2error", "", "cg_time", "cg_its",
                "cg_reduction");
    for (std::size_t i = 0; i < table.size(); ++i)
      {
        const Row &r = table[i];
        char       rate_f[16] = "-", rate_c[16] = "-";
        if (i > 0)
          {
            const double dc = std::log2((double)r.cells / table[i - 1].cells) / 3.;
            std::snprintf(rate_f, sizeof(rate_f), "%.2f", std::log2(table[i - 1].fmg_error / r.fmg_error) / dc);
            std::snprintf(rate_c, sizeof(rate_c), "%.2f", std::log2(table[i - 1].cg_error / r.cg_error) / dc);
          }
        std::printf("%-9zu %-11zu %-10.3e %-10.3e %-10.3e %-10.3e %-5s %-10.3e %-10.3e %-5s %-10.3e %-6u %-12.3e\n", r.cells,
                    r.dofs, r.mv_outer, r.mv_inner, r.reduction, r.fmg_error, rate_f, r.fmg_time, r.cg_error, rate_c,
                    r.cg_time, r.cg_its, r.cg_reduction);
      }
    std::cout << std::endl;
  }

  template <typename vcycle_number>
  void dispatch(unsigned int degree, std::size_t minsize, std::size_t maxsize, unsigned int c, unsigned int pre,
                unsigned int post, bool dbl)
  {
    switch (degree) // LaplaceRunTime<dim,1,9> (program.cc:614-643)
      {
        case 1: run<1, vcycle_number>(minsize, maxsize, c, pre, post, dbl); break;
        case 2: run<2, vcycle_number>(minsize, maxsize, c, pre, post, dbl); break;
        case 3: run<3, vcycle_number>(minsize, maxsize, c, pre, post, dbl); break;
        case 4: run<4, vcycle_number>(minsize, maxsize, c, pre, post, dbl); break;
        case 5: run<5, vcycle_number>(minsize, maxsize, c, pre, post, dbl); break;
        case 6: run<6, vcycle_number>(minsize, maxsize, c, pre, post, dbl); break;
        case 7: run<7, vcycle_number>(minsize, maxsize, c, pre, post, dbl); break;
        case 8: run<8, vcycle_number>(minsize, maxsize, c, pre, post, dbl); break;
        case 9: run<9, vcycle_number>(minsize, maxsize, c, pre, post, dbl); break;
        default: break; // degrees outside [1,9] do no work (program.cc:65-66)
      }
  }
} // namespace

int main(int argc, char *argv[])
{
  try
    {
      unsigned int degree = 0, n_mg_cycles = 1, n_pre_smooth = 3, n_post_smooth = 3; // program.cc:666-672
      std::size_t  maxsize = static_cast<std::size_t>(-1), minsize = 1;
      bool         use_doubling_mesh = true, vcycle_f64 = false; // program.cc:672
      if (argc == 1)
        {
          std::cout << "Expected at least one argument." << std::endl
                    << "Usage:" << std::endl
                    << "./poisson_cube degree minsize maxsize n_mg_cycles n_pre_smooth n_post_smooth doubling [f32|f64]"
                    << std::endl
                    << "The parameters degree to n_post_smooth are integers, "
                    << "the last selects between a square mesh or a doubling mesh" << std::endl;
          return 1;
        }
      if (argc > 1)
        degree = std::atoi(argv[1]);
      if (argc > 2)
        minsize = std::atoll(argv[2]);
      if (argc > 3)
        maxsize = std::atoll(argv[3]);
      if (argc > 4)
        n_mg_cycles = std::atoi(argv[4]);
      if (argc > 5)
        n_pre_smooth = std::atoi(argv[5]);
      if (argc > 6)
        n_post_smooth = std::atoi(argv[6]);
      if (argc > 7)
        use_doubling_mesh = argv[7][0] == 'd';
      if (argc > 8)
        vcycle_f64 = std::strcmp(argv[8], "f64") == 0;
      std::cout << "Settings of parameters: " << std::endl
                << "Number of MPI ranks:            " << 1 << " (one MI355X)" << std::endl
                << "Polynomial degree:              " << degree << std::endl
                << "Minimum size:                   " << minsize << std::endl
                << "Maximum size:                   " << maxsize << std::endl
                << "Number of MG cycles in V-cycle: " << n_mg_cycles << std::endl
                << "Number of pre-smoother iters:   " << n_pre_smooth << std::endl
                << "Number of post-smoother iters:  " << n_post_smooth << std::endl
                << "Use doubling mesh:              " << use_doubling_mesh << std::endl
                << "V-cycle number type:            " << (vcycle_f64 ? "double" : "float") << std::endl
                << std::endl;
      if (vcycle_f64)
        dispatch<double>(degree, minsize, maxsize, n_mg_cycles, n_pre_smooth, n_post_smooth, use_doubling_mesh);
      else
        dispatch<float>(degree, minsize, maxsize, n_mg_cycles, n_pre_smooth, n_post_smooth, use_doubling_mesh);
    }
  catch (std::exception &exc) // program.cc:717-727
    {
      std::cerr << std::endl
                << std::endl
                << "----------------------------------------------------" << std::endl;
      std::cerr << "Exception on processing: " << std::endl
                << exc.what() << std::endl
                << "Aborting!" << std::endl
                << "----------------------------------------------------" << std::endl;
      return 1;
    }
  return 0;
}
