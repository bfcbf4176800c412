"""Harness of BASELINE config 5's solver, mirroring poisson_dg/program.cc: FE_DGQHermite(p) on the cube
[-0.9, 1]^3, rhs 3 (3 pi)^2 prod sin(3 pi x_d), V-cycle-preconditioned CG with the DG level on top of the
FE_Q(p) multigrid (fp32 V-cycle inside the fp64 outer iteration, program.cc:72-73).

    python tools/poisson_dg.py [degree=3] [n_refine=5] [n_pre_smooth=3] [tolerance=1e-9] [--vcycle f32|f64]

Prints the reference's lines ("Time solve CG", "matvec time dp/sp ... DoFs/s", "L2 error with ndof = ...") and
the row of its convergence table (cells dofs mv_outer mv_inner cg_L2error cg_time cg_its cg_reduction,
program.cc:318-325).  Right-hand side and error norm are evaluated on the host (numpy), as the reference does
on the CPU; everything timed runs on the GPU."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigrid_amd as mg  # noqa: E402

WAVE = 3.0


def quad_points(cube, ijk, xq, h):
    """[cell, q = (k, j, i), 3] coordinates of the Gauss points"""
    n = xq.size
    ref = np.stack(np.meshgrid(xq, xq, xq, indexing="ij"), axis=-1)[..., ::-1].reshape(-1, 3)  # (k, j, i) order
    return -0.9 + h * (ijk[:, None, :] + ref[None, :, :])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("degree", nargs="?", type=int, default=3)
    ap.add_argument("n_refine", nargs="?", type=int, default=5)
    ap.add_argument("n_pre_smooth", nargs="?", type=int, default=3)
    ap.add_argument("tolerance", nargs="?", type=float, default=1e-9)
    ap.add_argument("--vcycle", choices=["f32", "f64"], default="f32")
    ap.add_argument("--basis", type=int, default=0)
    a = ap.parse_args()
    vnum = mg.F32 if a.vcycle == "f32" else mg.F64
    t0 = time.time()
    ctx = mg.Context(0)
    cube = mg.Cube(a.degree, 1, a.n_refine)
    solver = mg.DGMultigridSolver(ctx, cube, a.basis, a.n_pre_smooth, vnum)
    n = solver.m()
    nc = n // (a.degree + 1) ** 3
    print("Number of degrees of freedom: %d (%d cells, FE_DGQHermite(%d) on FE_Q(%d) multigrid, V-cycle in %s)"
          % (n, nc, a.degree, a.degree, a.vcycle))
    S, xq, wq = solver.matrix_dg.basis_1d()
    h = cube.cell_size(cube.max_level)
    S3 = np.kron(S, np.kron(S, S))
    w3 = np.kron(wq, np.kron(wq, wq)) * h ** 3
    x = quad_points(cube, solver.cell_ijk.astype(float), xq, h)
    u = np.prod(np.sin(np.pi * WAVE * x), axis=-1)
    rhs = ((3 * (np.pi * WAVE) ** 2 * u) * w3) @ S3          # program.cc:137-141, multigrid_solver_dg.h:243-262
    print("Time setup                    %.3f s   rhs_norm = %.6e" % (time.time() - t0, np.linalg.norm(rhs)))
    b, sol = ctx.vector(n, data=rhs.ravel()), ctx.vector(n)
    time_cg = 1e10
    for _ in range(4):                                          # program.cc:252-258
        ctx.sync()
        t = time.perf_counter()
        its, red = solver.solve_cg(b, sol, a.tolerance)
        ctx.sync()
        dt = time.perf_counter() - t
        time_cg = min(time_cg, dt)
        print("Time solve CG                 %.6f s   (%d iterations, reduction %.4e)" % (dt, its, red))
    uh = sol.download().reshape(nc, -1) @ S3.T
    l2 = np.sqrt(np.sum(w3 * (uh - u) ** 2) / (nc * h ** 3))    # multigrid_solver_dg.h:328-367
    A_dp, A_sp = solver.matrix_dg_dp, solver.matrix_dg
    best = {}
    for name, A, number in (("dp", A_dp, mg.F64), ("sp", A_sp, vnum)):
        v, w = ctx.vector(n, number, np.ones(n)), ctx.vector(n, number)
        n_mv = 200 if n < 10000000 else 50
        best[name] = 1e10
        for _ in range(5):
            ctx.sync()
            t = time.perf_counter()
            for _ in range(n_mv):
                A.vmult(w, v)
            ctx.sync()
            dt = (time.perf_counter() - t) / n_mv
            best[name] = min(best[name], dt)
            print("matvec time %s %.6e DoFs/s: %.5e" % (name, dt, n / dt))
        v.free(); w.free()
    print("Best timings for ndof = %d   mv %.6e    mv smooth %.6e   cg-mg %.6e" % (n, best["dp"], best["sp"], time_cg))
    print("L2 error with ndof = %d  %.6e" % (n, l2))
    print("cells dofs mv_outer mv_inner cg_L2error cg_time cg_its cg_reduction")
    print("%d %d %.4e %.4e %.4e %.4e %d %.4e  | %.3e DoFs/s solved per second of CG"
          % (nc, n, best["dp"], best["sp"], l2, time_cg, its, red, n / time_cg))
    solver.close()
    cube.close()
    ctx.close()


if __name__ == "__main__":
    main()
