"""PCG (V-cycle preconditioned) at full size: plain vector kernels vs the merged operations
(vmult_with_cg_update fused into the brick loop, vmult_with_residual_update around the V-cycle)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigrid_amd as mg
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 128
vnum = mg.F32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else mg.F64
ns, nr = cells, 0
while ns % 2 == 0 and ns > 1:
    ns //= 2; nr += 1
ctx = mg.Context(0)
cube = mg.Cube(4, ns, nr)
solver = mg.MultigridSolver(ctx, cube, 3, 3, 1, vnum)
for name, fn in (("plain", solver.solve_cg), ("fused", solver.solve_cg_fused)):
    fn(); ctx.sync()
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); its, red = fn(); ctx.sync(); best = min(best, time.perf_counter() - t)
    print("%s PCG: %d iterations, reduction %.4e, %.2f ms (%.2f ms per iteration), L2 error %.4e"
          % (name, its, red, best * 1e3, best * 1e3 / its, solver.compute_l2_error()))
