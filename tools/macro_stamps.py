"""Phase timeline of the macro-element brick kernel (diagnostic build: make -C multigrid_amd/csrc
MACROFLAGS=-DMGX_MACRO_STAMPS).  Prints median cycles per phase over the workgroups of the last
colour launch.  usage: macro_stamps.py [cells] [vmult|cheb|prolong] [degree]
(prolong: a V-cycle with Chebyshev degree 1, whose last finest-level launch is the prolongation form, mode 9)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multigrid_amd as mg
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 128
mode = sys.argv[2] if len(sys.argv) > 2 else "vmult"
degree = int(sys.argv[3]) if len(sys.argv) > 3 else 4
ns, nr = cells, 0
while ns % 2 == 0 and ns > 1:
    ns //= 2; nr += 1
# (prolong: the level below the finest one on the one-launch schedule, whose transfers are kernels of their own: the
# stamped launches of the fused transfer forms then all belong to the finest level)
ctx = mg.Context(0, options={"free_one_max": 8192} if mode == "prolong" else None)
cube = mg.Cube(degree, ns, nr)
l = cube.max_level
op = mg.LaplaceOperator.from_cube(ctx, cube, l)
x = ctx.vector(cube.n_dofs(l), data=cube.seeded_vector(l, 42))
y = ctx.vector(cube.n_dofs(l))
if mode == "vmult":
    for _ in range(5):
        op.vmult(y, x)
elif mode == "prolong":
    solver = mg.MultigridSolver(ctx, cube, 1, 1, 1, mg.F64)
    for _ in range(3):
        solver.vmult(y, x)
else:
    sm = mg.Chebyshev(op, 20., 3, 15)
    for _ in range(3):
        sm.step(y, x)
ctx.sync()
lib = mg._lib.load()
nb = 4096
buf = np.zeros((nb, 16), np.uint64)
# the second pipeline (mgx_macro2.hip) keeps stamps of its own: use them where it ran (MGX_STAMPS_V1=1: first pipeline)
v2 = hasattr(lib, "mgx_debug_read_stamps2") and not os.environ.get("MGX_STAMPS_V1")
if v2:
    rc = lib.mgx_debug_read_stamps2(buf.ctypes.data_as(ctypes.c_void_p), nb)
    v2 = rc == 0 and (buf[:, 0] != 0).any()
if not v2:
    rc = lib.mgx_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), nb)
assert rc == 0, rc
live = buf[:, 0] != 0
t = buf[live].astype(np.int64)
print("workgroups with stamps:", live.sum())
# stamps: 0 start, 1 first table staged, 2 first gather landed; per brick (values of the LAST brick of
# each workgroup): 3 loop top, 4 x done, 5 y done, 6 z done, 7 next table parked + next gather issued
# (absent for the last brick), 8 write-out done; 14 end (after vmcnt(0)); 15/13 = s_memrealtime at 0/14
clock = (t[:, 14] - t[:, 0]) / np.maximum(1, (t[:, 13] - t[:, 15])) * 100.0  # MHz
print("shader clock from s_memtime/s_memrealtime: median %.0f MHz (p10 %.0f, p90 %.0f)" %
      (np.median(clock), np.percentile(clock, 10), np.percentile(clock, 90)))
def rep(name, d):
    print("%-34s %8.0f %8.0f %8.0f cycles" % (name, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
rep("prologue: table", t[:, 1] - t[:, 0])
rep("prologue: gather + land", t[:, 2] - t[:, 1])
# fourth brick of each workgroup (steady state): 3 loop top, 4 x, 5 y, 6 z, 7 tables parked + next gather
# issued, 8 write-out done, 9 next gather landed in U, 10 barrier, 11 next loop top
if v2:
    # second pipeline: 3 loop top, 4 requests issued + barrier, 5 x done, 6 y done, 7 z done, 8 write-out done, 9 barrier,
    # 10 gather landed + tables stored, 11 next loop top
    print("(second pipeline)")
    rep("brick 4: issue gather / partials", t[:, 4] - t[:, 3])
    rep("brick 4: x sweep", t[:, 5] - t[:, 4])
    rep("brick 4: y sweep", t[:, 6] - t[:, 5])
    rep("brick 4: z sweep", t[:, 7] - t[:, 6])
    rep("brick 4: write-out", t[:, 8] - t[:, 7])
    rep("brick 4: barrier", t[:, 9] - t[:, 8])
    rep("brick 4: land gather, store tables", t[:, 10] - t[:, 9])
    rep("brick 4: barrier", t[:, 11] - t[:, 10])
else:
    rep("brick 4: x sweep", t[:, 4] - t[:, 3])
    rep("brick 4: y sweep", t[:, 5] - t[:, 4])
    rep("brick 4: z sweep", t[:, 6] - t[:, 5])
    rep("brick 4: park tables, issue gather", t[:, 7] - t[:, 6])
    rep("brick 4: write-out", t[:, 8] - t[:, 7])
    rep("brick 4: barrier + land gather", t[:, 9] - t[:, 8])
    rep("brick 4: barrier", t[:, 10] - t[:, 9])
rep("brick 4: whole iteration", t[:, 11] - t[:, 3])
rep("whole workgroup", t[:, 14] - t[:, 0])
rt = t[:, 13] - t[:, 15]
print("whole workgroup by s_memrealtime: median %.1f us; launch span %.1f us" %
      (np.median(rt) / 100.0, (t[:, 13].max() - t[:, 15].min()) / 100.0))
# how the launch ends: start and end of every workgroup relative to the first start (s_memrealtime, 100 MHz)
st, en = (t[:, 15] - t[:, 15].min()) / 100.0, (t[:, 13] - t[:, 15].min()) / 100.0
pc = lambda a: " ".join("%6.1f" % np.percentile(a, q) for q in (0, 10, 50, 90, 100))   # noqa: E731
print("workgroup start  [us] min p10 p50 p90 max:", pc(st))
print("workgroup end    [us] min p10 p50 p90 max:", pc(en))
print("workgroup length [us] min p10 p50 p90 max:", pc(en - st), "  mean %.1f" % (en - st).mean())
