"""Brick form of the general operator (brick_general_kernel) against the per-cell kernel + ordered assembly on the
same mesh: maximum difference of the products, and the time of both.  usage: general_bricks_check.py <refinements> [6|12] [f32]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multigrid_amd as mg
nr = int(sys.argv[1]) if len(sys.argv) > 1 else 3
nc = int(sys.argv[2]) if len(sys.argv) > 2 else 6
num = mg.F32 if len(sys.argv) > 3 and sys.argv[3] == "f32" else mg.F64
res = {}
for name, opts in (("bricks", {"general_brick_min": 1}), ("cells", {"no_general_bricks": 1})):
    ctx = mg.Context(0, options=opts)
    cube = mg.Cube(4, n_refine=nr, shell=nc, problem="shell")
    l = cube.max_level
    n = cube.n_dofs(l)
    op = mg.LaplaceOperator.from_cube(ctx, cube, l, num)
    dt = np.float32 if num == mg.F32 else np.float64
    x, y = ctx.vector(n, num, cube.seeded_vector(l, 1).astype(dt)), ctx.vector(n, num)
    for _ in range(3):
        op.vmult(y, x)
    ctx.sync()
    t = time.perf_counter()
    for _ in range(20):
        op.vmult(y, x)
    ctx.sync()
    res[name] = (y.download().astype(np.float64), (time.perf_counter() - t) / 20)
    again = ctx.vector(n, num)
    op.vmult(again, x)
    assert np.array_equal(again.download(), y.download()), "not reproducible"
    print("%-6s %d cells %d DoFs: vmult %.3f ms = %.3e DoFs/s" % (name, cube.n_cells(l), n, 1e3 * res[name][1], n / res[name][1]))
a, b = res["bricks"][0], res["cells"][0]
print("max |bricks - cells| / max |cells| = %.3e" % (np.abs(a - b).max() / np.abs(b).max()))
d = np.abs(a - b)
bad = np.nonzero(d > 1e-9 * np.abs(b).max())[0]
print("differing DoFs: %d of %d; first %s" % (bad.size, a.size, bad[:12]))
if bad.size:
    print("bricks values", a[bad[:6]], "cells values", b[bad[:6]])
    print("zeros among the differing brick values: %d; ratio stats %s" % ((a[bad] == 0).sum(), np.percentile(a[bad] / np.where(b[bad] == 0, 1, b[bad]), [0, 25, 50, 75, 100])))
