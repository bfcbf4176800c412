"""the flow of tests/test_gpu_fullsize.py::test_production_path_equals_plain_path_at_scale with extra
comparisons: which of the two results moves when they differ?  usage: p ns nr"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import multigrid_amd as mg
FALLBACKS = {"MGX_NO_FUSED_RESTRICT": "1", "MGX_NO_FUSED_INIT": "1", "MGX_TRANSFER_V1": "1",
             "MGX_RESTRICT_ATOMIC": "1", "MGX_BRICK_WIDE_MAX": "0", "MGX_NO_GRAPH": "1", "MGX_BRICK_FORM": "cells",
             "MGX_NO_DIAG_TABLE": "1", "MGX_NO_FUSED_PROLONG": "1"}
p, ns, nr = map(int, sys.argv[1:4])
ctx = mg.Context(0)
cube = mg.Cube(p, ns, nr)
l = cube.max_level; n = cube.n_dofs(l)
x = ctx.vector(n, data=cube.seeded_vector(l, 11))
solver = mg.MultigridSolver(ctx, cube, 3, 3, 1, mg.F64)
a = ctx.vector(n)
solver.vmult(a, x)
solver.vmult(a, x)
for k, v in FALLBACKS.items():
    os.environ[k] = v
ctx2 = mg.Context(0)
plain = mg.MultigridSolver(ctx2, cube, 3, 3, 1, mg.F64)
b = ctx.vector(n)
plain.vmult(b, x)
ctx2.sync(); ctx.sync()
A, B = a.download(), b.download()
a2, b2 = ctx.vector(n), ctx.vector(n)
solver.vmult(a2, x); ctx.sync()
plain.vmult(b2, x); ctx2.sync()
A2, B2 = a2.download(), b2.download()
nb = np.linalg.norm(B2)
lam = [solver.smoother(lv).info()["lambda_max"] for lv in range(cube.n_levels)]
laml = [plain.smoother(lv).info()["lambda_max"] for lv in range(cube.n_levels)]
print("a-b %.2e  a-a2 %.2e  b-b2 %.2e  a2-b2 %.2e | lambda diffs %s" % (
    np.linalg.norm(A - B) / nb, np.linalg.norm(A - A2) / nb, np.linalg.norm(B - B2) / nb, np.linalg.norm(A2 - B2) / nb,
    " ".join("%.0e" % abs(u / v - 1) for u, v in zip(lam, laml))), flush=True)
