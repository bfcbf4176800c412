"""Runs N finest-level restrictions and prolongations -- profiling target for the transfer kernels."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigrid_amd as mg
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ns, nr = cells, 0
while ns % 2 == 0 and ns > 1:
    ns //= 2; nr += 1
ctx = mg.Context(0)
cube = mg.Cube(4, ns, nr)
l = cube.max_level
opc = mg.LaplaceOperator.from_cube(ctx, cube, l - 1)
opf = mg.LaplaceOperator.from_cube(ctx, cube, l)
tr = mg.Transfer(opc, opf, cube.children(l), cube.prolong_1d())
xf = ctx.vector(cube.n_dofs(l), data=cube.seeded_vector(l, 1))
xc = ctx.vector(cube.n_dofs(l - 1), data=cube.seeded_vector(l - 1, 2))
def timed(fn):
    fn(); ctx.sync(); t = time.perf_counter()
    for _ in range(n): fn()
    ctx.sync(); return (time.perf_counter() - t) / n * 1e3
print("restrict_and_add ms %.3f" % timed(lambda: tr.restrict_and_add(xc, xf, with_constraints=True)))
print("prolongate_and_add ms %.3f" % timed(lambda: tr.prolongate_and_add(xf, xc, with_constraints=True)))
print("fine dofs", cube.n_dofs(l))
