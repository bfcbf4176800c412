"""Times the finest-level fp64 matvec / Chebyshev step (host timer around synchronised loops)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigrid_amd as mg
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 128
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ns, nr = cells, 0
while ns % 2 == 0 and ns > 1:
    ns //= 2; nr += 1
ctx = mg.Context(0)
cube = mg.Cube(deg, ns, nr)
l = cube.max_level
op = mg.LaplaceOperator.from_cube(ctx, cube, l)
x = ctx.vector(cube.n_dofs(l), data=cube.seeded_vector(l, 42))
y = ctx.vector(cube.n_dofs(l))
def timed(fn, n=10):
    fn(); ctx.sync(); t = time.perf_counter()
    for _ in range(n): fn()
    ctx.sync(); return (time.perf_counter() - t) / n * 1e3
print("ablate", os.environ.get("MGX_BRICK_ABLATE"), "vmult ms %.3f" % timed(lambda: op.vmult(y, x)))
if not os.environ.get("MGX_BRICK_ABLATE"):
    sm = mg.Chebyshev(op, 20., 3, 15)
    print("cheb step (3 fused its) ms %.3f" % timed(lambda: sm.step(y, x), 5))
    print("cheb vmult (2 fused its) ms %.3f" % timed(lambda: sm.vmult(y, x), 5))
