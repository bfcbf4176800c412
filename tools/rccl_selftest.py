"""Single-GPU exercise of the native RCCL transport (mgx_context_set_rccl): a one-rank communicator
whose only "neighbour" is the rank itself -- ncclSend/ncclRecv to self inside a group, issued by
the library on its stream -- checked against the expected sum, plus ncclAllReduce through a dot
product.  (Real peers need several GPUs; this covers id plumbing, dlopen, group calls, stream order.)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multigrid_amd as mg
from multigrid_amd import _lib

ctx = mg.Context(0, options={"rccl_selftest": 1})
lib = ctx.lib
buf = (C.c_uint8 * 128)()
mg.check(lib.mgx_rccl_unique_id(buf))
mg.check(lib.mgx_context_set_rccl(ctx.h, 0, 1, buf))
cube = mg.Cube(3, 1, 2)
l = cube.max_level
n = cube.n_dofs(l)
d = cube.operator_desc(l)
# fake plan: two "neighbours", both this rank, over disjoint unconstrained index lists
nfree = n - cube.n_constrained(l)
i0 = np.arange(5, 205, dtype=np.uint32)
i1 = np.arange(300, 1000, 3, dtype=np.uint32)
assert i1.max() < nfree
shared = np.concatenate([i0, i1]).astype(np.uint32)
ex = _lib.ExchangeDesc()
ranks = (C.c_int * 2)(0, 0)
counts = (C.c_uint32 * 2)(i0.size, i1.size)
idxp = (_lib.u32p * 2)(i0.ctypes.data_as(_lib.u32p), i1.ctypes.data_as(_lib.u32p))
ex.plan_id, ex.n_neighbors = 1, 2
ex.neighbor_rank = C.cast(ranks, C.POINTER(C.c_int))
ex.count = C.cast(counts, _lib.u32p)
ex.index = C.cast(idxp, C.POINTER(_lib.u32p))
ex.shared, ex.n_shared = shared.ctypes.data_as(_lib.u32p), shared.size
ex.not_owned, ex.n_not_owned = None, 0
ex.send_buf, ex.recv_buf = None, None
d.exchange = C.pointer(ex)
op = mg.LaplaceOperator(ctx, d)
v = cube.seeded_vector(l, 7)
x = ctx.vector(n, data=v)
for rep in range(3):
    x.upload(v)
    mg.check(lib.mgx_exchange_add(op.h, x.ptr))
    got = x.download()
    ref = v.copy()
    ref[shared] *= 2.0  # own value + the copy that came back from "the neighbour"
    assert np.array_equal(got, ref), np.abs(got - ref).max()
d1 = ctx.dot(x, x)   # local dot + ncclAllReduce over the one rank
assert abs(d1 - np.dot(ref, ref)) <= 1e-12 * d1, (d1, np.dot(ref, ref))
# a V-cycle-like sequence of exchanges back to back on the stream (no host sync in between)
y = ctx.vector(n)
for rep in range(20):
    op.vmult(y, x)
ctx.sync()
print("rccl selftest ok: send/recv to self inside a group on the solver stream, allreduce; dot %.12e" % d1)
