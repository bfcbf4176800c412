"""multigrid_amd -- MI355X-native matrix-free geometric multigrid for the Laplace operator.

Host-side mirror (Python, for tests and bench.py) of the reference's operator / solver interface
for the poisson_cube path on top of the C ABI in include/mgx.h:

    LaplaceOperator  <-> multigrid::LaplaceOperator   common/laplace_operator.h:56-164
    MultigridSolver  <-> multigrid::MultigridSolver   common/multigrid_solver.h:96-782
    DGLaplaceOperator <-> multigrid::LaplaceOperatorCompactCombine + JacobiTransformed
                                                       common/laplace_operator_dg.h:350-2256
    Cube             <-> the deal.II mesh / DoFHandler / MatrixFree data of poisson_cube/program.cc

All numerical work runs in hand-written HIP kernels inside libmgx.so; nothing here computes.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import F32, F64, INVALID_INDEX, MgxError, check

__all__ = ["Context", "DeviceVector", "Cube", "LaplaceOperator", "Chebyshev", "Transfer", "MultigridSolver",
           "Communicator", "process_grid", "F32", "F64", "INVALID_INDEX", "MgxError", "DGLaplaceOperator", "dg_cheby_mesh",
           "dg_box_neighbours", "dg_box_partition", "dg_partition", "DG_HERMITE", "DG_GAUSS_LOBATTO", "DG_GAUSS", "DGMultigridSolver",
           "spawn_ranks"]


def spawn_ranks(script, argv, n, one_gpu=False, grace=30.0):
    """Start n ranks of `script` (fresh child processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*; the
    calling process never touches the GPU) and wait for them.  Once a rank has failed the others get `grace`
    seconds -- they may sit in a collective the failed rank never joins -- and are then ended (exactly the
    processes started here).  Returns the exit codes."""
    import os
    import socket
    import subprocess
    import sys
    import time
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0" if one_gpu else str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env))
    failed_at = None
    while any(p.poll() is None for p in procs):
        time.sleep(0.5)
        if failed_at is None and any(p.poll() not in (None, 0) for p in procs):
            failed_at = time.time()
        if failed_at is not None and time.time() - failed_at > grace:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10.0)
                except subprocess.TimeoutExpired:
                    p.kill()
    return [p.wait() for p in procs]


def process_grid(size):
    """1 / 2x1x1 / 2x2x1 / 2x2x2; generally the most cubic power-of-two grid"""
    procs = [1, 1, 1]
    d = 0
    while size > 1:
        if size % 2:
            raise ValueError("number of ranks must be a power of two")
        procs[d] *= 2
        size //= 2
        d = (d + 1) % 3
    return tuple(procs)


class Communicator:
    """Transport of the interface exchange and of the scalar reductions (mgx_comm_desc) on top of
    torch.distributed: backend "nccl" (= RCCL over xGMI, one process per GPU) or "gloo" (CPU
    processes; used by the tests, also with several processes sharing one GPU).

    exchange(plan): the library has packed its send buffers (device memory); they are delivered
    with one batch of point-to-point operations (at most 7 neighbours in a 2x2x2 process grid,
    each a direct xGMI peer)."""

    def __init__(self, ctx, dist, device_transport=None, native=None):
        """native: also create an RCCL communicator inside the library (mgx_context_set_rccl; the
        unique id travels over `dist`).  It stays switched off until verify_and_enable_native() has
        compared one exchange and one reduction with the callback transport on every rank.
        Default: on for the nccl backend unless MGX_NATIVE_RCCL=0."""
        self.ctx, self.dist = ctx, dist
        self.rank, self.size = dist.get_rank(), dist.get_world_size()
        self.device_transport = (dist.get_backend() == "nccl") if device_transport is None else device_transport
        self.native_ready = False
        self.native_enabled = False
        self._native_wanted = (self.device_transport and os.environ.get("MGX_NATIVE_RCCL", "1") != "0") \
            if native is None else native
        self._tensors = {}   # device pointer -> torch CUDA tensor backing an exchange buffer
        self._p2p = {}       # plan_id -> cached P2POp list
        self._ex = _lib.EXCHANGE_FN(self._exchange)
        self._ar = _lib.ALLREDUCE_FN(self._allreduce)
        self._al = _lib.ALLOC_FN(self._alloc) if self.device_transport else _lib.ALLOC_FN()
        self.desc = _lib.CommDesc(self.rank, self.size, None, self._ex, self._ar, self._al)
        check(ctx.lib.mgx_context_set_comm(ctx.h, C.byref(self.desc)))
        ctx._comm = self  # keep the callbacks alive
        if self._native_wanted:
            self._setup_native()

    def _setup_native(self):
        """collective: every rank must arrive here; a failure on any rank disables it on all"""
        import sys
        import torch
        lib, h = self.ctx.lib, self.ctx.h
        buf = (C.c_uint8 * 128)()
        ok = 1
        if self.rank == 0 and lib.mgx_rccl_unique_id(buf) != 0:
            ok = 0
        dev = "cuda" if self.dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor(list(buf) + [ok], dtype=torch.uint8, device=dev)
        self.dist.broadcast(t, 0)
        vals = t.cpu().tolist()
        if vals[128] == 0:
            print("mgx Communicator: librccl unavailable, callback transport stays", file=sys.stderr, flush=True)
            return
        idb = (C.c_uint8 * 128)(*vals[:128])
        rc = lib.mgx_context_set_rccl(h, self.rank, self.size, idb)
        if rc == 0:
            rc = lib.mgx_context_use_rccl(h, 0)  # off until verified
        flag = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=dev)
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN)
        self.native_ready = int(flag.item()) == 1
        if not self.native_ready:
            print("mgx Communicator: RCCL communicator not created on every rank (%s), callback transport stays"
                  % lib.mgx_last_error().decode(), file=sys.stderr, flush=True)

    def verify_and_enable_native(self, op, n_dofs, number=F64):
        """One interface exchange and one dot product of a test vector through both transports;
        the native one is switched on only if every rank finds them bitwise identical."""
        if not self.native_ready:
            return False
        import sys
        import torch
        lib, h = self.ctx.lib, self.ctx.h
        rng = np.random.default_rng(1234 + self.rank)
        v = rng.uniform(-1, 1, n_dofs).astype(_DT[number])
        a, b = self.ctx.vector(n_dofs, number, v), self.ctx.vector(n_dofs, number, v)
        good = True
        try:
            check(lib.mgx_exchange_add(op.h, a.ptr))        # callbacks
            da = self.ctx.dot(a, a)
            check(lib.mgx_context_use_rccl(h, 1))
            check(lib.mgx_exchange_add(op.h, b.ptr))        # native
            db = self.ctx.dot(b, b)
            good = np.array_equal(a.download(), b.download()) and abs(da - db) <= 1e-12 * abs(da)
        except Exception as e:  # noqa: BLE001
            print("mgx Communicator: native RCCL check failed:", repr(e), file=sys.stderr, flush=True)
            good = False
        lib.mgx_context_use_rccl(h, 0)
        dev = "cuda" if self.dist.get_backend() == "nccl" else "cpu"
        flag = torch.tensor([1 if good else 0], dtype=torch.int32, device=dev)
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN)
        self.native_enabled = int(flag.item()) == 1
        if self.native_enabled:
            check(lib.mgx_context_use_rccl(h, 1))
        elif self.rank == 0:
            print("mgx Communicator: native RCCL transport disagrees with the callback transport, not used",
                  file=sys.stderr, flush=True)
        return self.native_enabled

    def _alloc(self, user, nbytes):
        """exchange buffers as torch CUDA tensors: RCCL sends from / receives into them directly"""
        try:
            import torch
            t = torch.zeros(max(int(nbytes), 8), dtype=torch.uint8, device="cuda")
            self._tensors[t.data_ptr()] = t
            return t.data_ptr()
        except Exception as e:
            import sys
            print("mgx Communicator.alloc failed:", repr(e), file=sys.stderr, flush=True)
            return None

    def _device_tensor(self, ptr, nbytes):
        """torch view of `nbytes` of device memory at `ptr`: the tensor the alloc callback handed out, or -- for
        buffers the library owns (the DG ghost exchange sends from hipMalloc'd pack buffers and receives straight
        into the ghost part of the vector it was given, mgx_dg.hip) -- a zero-copy wrapper of the raw pointer"""
        import torch
        t = self._tensors.get(ptr)
        if t is not None:
            return t[:nbytes]

        class _Raw:  # __cuda_array_interface__ v2: torch wraps the memory, it does not own it
            pass
        raw = _Raw()
        raw.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
        return torch.as_tensor(raw, device="cuda")

    def _exchange(self, user, plan_id, number, n_neighbors, ranks, counts, send, recv):
        try:
            import torch
            dt = torch.float64 if number == F64 else torch.float32
            es = 8 if number == F64 else 4
            if self.device_transport:
                # the operations are cached per plan AND buffer set: one plan serves many vectors when the
                # library receives in place (DG ghosts), each with its own receive pointers
                key = (plan_id, tuple(send[k] for k in range(n_neighbors)), tuple(recv[k] for k in range(n_neighbors)))
                ops = self._p2p.get(key)
                if ops is None:
                    ops = []
                    for k in range(n_neighbors):
                        cnt = counts[k]
                        st = self._device_tensor(send[k], cnt * es).view(dt)
                        rt = self._device_tensor(recv[k], cnt * es).view(dt)
                        ops.append(self.dist.P2POp(self.dist.isend, st, ranks[k]))
                        ops.append(self.dist.P2POp(self.dist.irecv, rt, ranks[k]))
                    if len(self._p2p) > 256:
                        self._p2p.clear()
                    self._p2p[key] = ops
                if ops:
                    for req in self.dist.batch_isend_irecv(ops):
                        req.wait()
                    torch.cuda.synchronize()
                return 0
            lib, h = self.ctx.lib, self.ctx.h
            ops, bufs = [], []
            for k in range(n_neighbors):
                rk, cnt = ranks[k], counts[k]
                st = torch.empty(cnt, dtype=dt)
                rt = torch.empty(cnt, dtype=dt)
                check(lib.mgx_download(h, C.c_void_p(st.data_ptr()), C.c_void_p(send[k]), cnt * es))
                bufs.append((rt, recv[k], cnt))
                ops.append(self.dist.P2POp(self.dist.isend, st, rk))
                ops.append(self.dist.P2POp(self.dist.irecv, rt, rk))
            if ops:
                for req in self.dist.batch_isend_irecv(ops):
                    req.wait()
            for (rt, rbuf, cnt) in bufs:
                check(lib.mgx_upload(h, C.c_void_p(rbuf), C.c_void_p(rt.data_ptr()), cnt * es))
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            import sys
            print("mgx Communicator.exchange failed:", repr(e), file=sys.stderr, flush=True)
            return 1

    def _allreduce(self, user, values, count):
        try:
            import torch
            a = np.ctypeslib.as_array(values, shape=(count,))
            t = torch.from_numpy(a.copy())
            if self.device_transport:
                t = t.cuda()
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
            a[:] = t.cpu().numpy()
            return 0
        except Exception as e:
            import sys
            print("mgx Communicator.allreduce failed:", repr(e), file=sys.stderr, flush=True)
            return 1

    def allreduce(self, values):
        a = np.ascontiguousarray(values, dtype=np.float64)
        if self._allreduce(None, a.ctypes.data_as(_lib.f64p), a.size) != 0:
            raise RuntimeError("allreduce failed")
        return a

_DT = {F32: np.float32, F64: np.float64}


class Context:
    """HIP device + stream (mgx_context_t)."""

    def __init__(self, device=0, options=None):
        """options: {name: value} for mgx_context_set_option (code-path selectors and thresholds; include/mgx.h)"""
        self.lib = _lib.load()
        h = C.c_void_p()
        check(self.lib.mgx_context_create(C.byref(h), device))
        self.h = h
        self.device = device
        for name, value in (options or {}).items():
            self.set_option(name, value)

    def set_option(self, name, value):
        check(self.lib.mgx_context_set_option(self.h, name.encode(), float(value)))

    def sync(self):
        check(self.lib.mgx_sync(self.h))

    @property
    def stream(self):
        return self.lib.mgx_context_stream(self.h)

    def profile_enable(self, on=True):
        check(self.lib.mgx_profile_enable(self.h, int(on)))

    def range(self, name):
        """context manager: a profiler range (roctx) with one of the reference's LIKWID region names --
        "fmg_solver", "cg_solver", "matvec", "matvec_sp" (poisson_cube/program.cc:282-375); no-op unless the
        context option "roctx" is set"""
        ctx = self

        class _Range:
            def __enter__(self):
                check(ctx.lib.mgx_range_push(ctx.h, name.encode()))

            def __exit__(self, *exc):
                check(ctx.lib.mgx_range_pop(ctx.h))
                return False
        return _Range()

    def profile_read(self, form):
        """(kernel launches, summed duration in ms) of one form of the profiled cell loop:
        0 plain vmult, 1 residual, 2 fused Chebyshev iteration, 3 first step, 4 zero x_old; resets"""
        n, ms = C.c_uint64(), C.c_double()
        check(self.lib.mgx_profile_read(self.h, form, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def close(self):
        if getattr(self, "h", None):
            self.lib.mgx_context_destroy(self.h)
            self.h = None

    def vector(self, n, number=F64, data=None):
        return DeviceVector(self, n, number, data)

    def dot(self, x, y):
        r = C.c_double()
        check(self.lib.mgx_dot(self.h, x.number, x.ptr, y.ptr, x.n, C.byref(r)))
        return r.value

    def l2_norm(self, x):
        r = C.c_double()
        check(self.lib.mgx_l2_norm(self.h, x.number, x.ptr, x.n, C.byref(r)))
        return r.value


class DeviceVector:
    """Device storage of one level vector (LinearAlgebra::distributed::Vector<number>)."""

    def __init__(self, ctx, n, number=F64, data=None, ptr=None):
        self.ctx, self.n, self.number = ctx, int(n), number
        self.owned = ptr is None
        if ptr is None:
            p = C.c_void_p()
            check(ctx.lib.mgx_malloc(ctx.h, C.byref(p), self.nbytes))
            self.ptr = p
            if data is None:
                self.zero()
        else:
            self.ptr = C.c_void_p(ptr) if not isinstance(ptr, C.c_void_p) else ptr
        if data is not None:
            self.upload(data)

    @property
    def nbytes(self):
        return self.n * (8 if self.number == F64 else 4)

    def zero(self):
        check(self.ctx.lib.mgx_memset_zero(self.ctx.h, self.ptr, self.nbytes))

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=_DT[self.number])
        assert a.size == self.n
        check(self.ctx.lib.mgx_upload(self.ctx.h, self.ptr, a.ctypes.data_as(C.c_void_p), self.nbytes))

    def download(self):
        a = np.empty(self.n, dtype=_DT[self.number])
        check(self.ctx.lib.mgx_download(self.ctx.h, a.ctypes.data_as(C.c_void_p), self.ptr, self.nbytes))
        return a

    def free(self):
        if self.owned and self.ptr:
            self.ctx.lib.mgx_free(self.ctx.h, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Cube:
    """Host-side discretisation of poisson_cube (include/mgx_cube.h): what deal.II supplies."""

    NUMBERING = {"brick": 0, "cell": 1}
    GEOMETRY = {"cartesian": 0, "sheared": 1, "shell_sector": 2}
    PROBLEM = {"cube": 0, "shell": 1}

    def __init__(self, degree, n_subdiv=1, n_refine=3, box=None, procs=(1, 1, 1), rank=0, numbering="brick",
                 origin=-1.0, h0=1.9, geometry="cartesian", problem="cube", shell=None):
        """box=None: the square mesh [-0.9,1]^3 with n_subdiv coarse cells per direction.
        box=(sx,sy,sz): a box of sx x sy x sz cubic coarse cells of size h0 from (origin,)*3,
        optionally distributed over the process grid `procs`; this rank owns box[d]/procs[d] coarse
        cells per direction on every level.  The defaults (size 1.9 from -1) are the reference's
        doubling-mesh family (program.cc:509-529: one coarse cube per rank = weak scaling);
        box=(n,n,n), origin=-0.9, h0=1.9/n is the square mesh of poisson_cube with n_subdiv = n,
        block-split over the ranks (strong scaling of one problem).
        geometry / problem (box form only): mapped meshes and the variable coefficient of
        poisson_shell, MGX_CUBE_GEOMETRY_* / MGX_CUBE_PROBLEM_* in mgx_cube.h.
        numbering: "brick" (default, grouped for the device cell loop) or "cell" (the
        plain first-touch order), MGX_CUBE_NUMBERING_* in mgx_cube.h.
        shell=6 | 12: the mesh of poisson_shell instead, GridGenerator::hyper_shell(0, 0.5, 1.0, shell) refined
        n_refine times (mgx_cube_create_shell); problem "shell" (default there) or "cube"; procs=(n,1,1), rank=r:
        the coarse cells distributed over n <= `shell` ranks (contiguous shares, one cell more on some ranks where n
        does not divide them: 12 cells on 8 ranks)."""
        self.lib = _lib.load()
        h = C.c_void_p()
        num = self.NUMBERING[numbering]
        self.shell = shell
        if shell is not None:
            self.box_desc = None
            self.shell_desc = dict(shell=int(shell), problem=problem)
            n_ranks = int(procs[0]) * int(procs[1]) * int(procs[2])
            check(self.lib.mgx_cube_create_shell_ranks(degree, int(shell), n_refine, self.PROBLEM[problem], n_ranks, rank,
                                                       C.byref(h)))
            self.h = h
            self.rank, self.size = rank, n_ranks
            self.degree = degree
            self.n_levels = self.lib.mgx_cube_n_levels(h)
            self.max_level = self.n_levels - 1
            # level i of this object is level i + level_offset of the whole mesh (1 where the cells of level 1 are
            # dealt out to the ranks: 8 ranks hold 6 of 48 / 12 of 96 cells each)
            self.level_offset = self.lib.mgx_cube_level_offset(h)
            return
        self.level_offset = 0
        self.box_desc = None if box is None else dict(box=tuple(box), origin=origin, h0=h0, geometry=geometry,
                                                        problem=problem, numbering=numbering)
        if box is None:
            check(self.lib.mgx_cube_create_numbered(degree, n_subdiv, n_refine, num, C.byref(h)))
        else:
            d = _lib.CubeBoxDesc(degree, n_refine, (C.c_int * 3)(*box), origin, h0, (C.c_int * 3)(*procs), rank, num,
                                 self.GEOMETRY[geometry], self.PROBLEM[problem])
            check(self.lib.mgx_cube_create_box(C.byref(d), C.byref(h)))
        self.h = h
        self.rank, self.size = self.lib.mgx_cube_rank(h), self.lib.mgx_cube_size(h)
        self.degree = degree
        self.n_levels = self.lib.mgx_cube_n_levels(h)
        self.max_level = self.n_levels - 1

    def close(self):
        if getattr(self, "h", None):
            self.lib.mgx_cube_destroy(self.h)
            self.h = None

    def n_cells(self, l):
        return self.lib.mgx_cube_n_cells(self.h, l)

    def n_dofs(self, l):
        return self.lib.mgx_cube_n_dofs(self.h, l)

    def n_constrained(self, l):
        return self.lib.mgx_cube_n_constrained(self.h, l)

    def cells_per_dim(self, l):
        return self.lib.mgx_cube_cells_per_dim(self.h, l)

    def cell_size(self, l):
        return self.lib.mgx_cube_cell_size(self.h, l)

    def _arr(self, fn, shape, *args):
        f = getattr(self.lib, fn)
        p = f(self.h, *args)
        n = int(np.prod(shape))
        if n == 0:
            return np.zeros(shape, dtype=np.dtype(f.restype._type_))
        return np.ctypeslib.as_array(p, shape=(n,)).reshape(shape).copy()

    def idx27(self, l):
        return self._arr("mgx_cube_idx27", (self.n_cells(l), 27), l)

    def idx27_plain(self, l):
        return self._arr("mgx_cube_idx27_plain", (self.n_cells(l), 27), l)

    def constrained(self, l):
        return self._arr("mgx_cube_constrained", (self.n_constrained(l),), l)

    def coef_q(self, l):
        """[n_cells, 6, (p+1)^3] merged coefficient of a mapped level (None on the Cartesian cube)"""
        ptr = self.lib.mgx_cube_coef_q(self.h, l)
        if not ptr:
            return None
        return np.ctypeslib.as_array(ptr, shape=(self.n_cells(l), 6, (self.degree + 1) ** 3)).copy()

    def children(self, l):
        return self._arr("mgx_cube_children", (self.n_cells(l - 1), 8), l)

    def cell_nodes(self, l):
        """multi-block meshes: physical Gauss-Lobatto points of every cell, [n_cells, 3, (p+1)^3]"""
        out = np.empty((self.n_cells(l), 3, (self.degree + 1) ** 3))
        check(self.lib.mgx_cube_cell_nodes(self.h, l, out.ctypes.data_as(_lib.f64p)))
        return out

    def entity_multiplicity(self, l):
        """multi-block meshes: cells around each of the 27 entities of every cell, [n_cells, 27]"""
        ptr = self.lib.mgx_cube_entity_multiplicity(self.h, l)
        if not ptr:
            return None
        return np.ctypeslib.as_array(ptr, shape=(self.n_cells(l) * 27,)).reshape(-1, 27).copy()

    def cell_coords(self, l):
        return self._arr("mgx_cube_cell_coords", (self.n_cells(l), 3), l)

    def dof_grid(self, l):
        return self._arr("mgx_cube_dof_grid", (self.n_dofs(l),), l)

    def shape_values(self):
        n = self.degree + 1
        return self._arr("mgx_cube_shape_values", (n, n))

    def colloc_grad(self):
        n = self.degree + 1
        return self._arr("mgx_cube_colloc_grad", (n, n))

    def qweights(self):
        return self._arr("mgx_cube_qweights", (self.degree + 1,))

    def qpoints(self):
        return self._arr("mgx_cube_qpoints", (self.degree + 1,))

    def gll(self):
        return self._arr("mgx_cube_gll", (self.degree + 1,))

    def prolong_1d(self):
        n = self.degree + 1
        return self._arr("mgx_cube_prolong_1d", (2 * n - 1, n))

    def rhs(self, l):
        return self._arr("mgx_cube_rhs", (self.n_dofs(l),), l)

    def rhs_quadrature(self, l):
        """f(x_q) JxW_q, [n_cells, (p+1)^3]: the integrand of the right-hand side (input of compute_residual)"""
        out = np.empty((self.n_cells(l), (self.degree + 1) ** 3))
        check(self.lib.mgx_cube_rhs_quadrature(self.h, l, out.ctypes.data_as(_lib.f64p)))
        return out

    def bc(self, l):
        n = self.lib.mgx_cube_bc_count(self.h, l)
        return self._arr("mgx_cube_bc_index", (n,), l), self._arr("mgx_cube_bc_value", (n,), l)

    def neighbors(self, l):
        """[(rank, index array)] of the interface exchange on level l (ascending rank)"""
        out = []
        for k in range(self.lib.mgx_cube_n_neighbors(self.h, l)):
            n = self.lib.mgx_cube_neighbor_count(self.h, l, k)
            out.append((self.lib.mgx_cube_neighbor_rank(self.h, l, k),
                        self._arr("mgx_cube_neighbor_index", (n,), l, k)))
        return out

    def shared(self, l):
        return self._arr("mgx_cube_shared", (self.lib.mgx_cube_n_shared(self.h, l),), l)

    def not_owned(self, l):
        return self._arr("mgx_cube_not_owned", (self.lib.mgx_cube_n_not_owned(self.h, l),), l)

    def cells_per_dim3(self, l):
        a, b = (C.c_uint32 * 3)(), (C.c_uint32 * 3)()
        self.lib.mgx_cube_cells_per_dim3(self.h, l, C.byref(a), C.byref(b))
        return tuple(a), tuple(b)

    def l2_error_parts(self, l, solution):
        s = np.ascontiguousarray(solution, dtype=np.float64)
        e, v = C.c_double(), C.c_double()
        self.lib.mgx_cube_l2_error_parts(self.h, l, s.ctypes.data_as(_lib.f64p), C.byref(e), C.byref(v))
        return e.value, v.value

    def operator_desc(self, l, number=F64):
        d = _lib.OperatorDesc()
        check(self.lib.mgx_cube_operator_desc(self.h, l, number, C.byref(d)))
        return d

    def l2_error(self, l, solution):
        s = np.ascontiguousarray(solution, dtype=np.float64)
        assert s.size == self.n_dofs(l)
        return self.lib.mgx_cube_l2_error(self.h, l, s.ctypes.data_as(_lib.f64p))

    def seeded_vector(self, l, seed=42):
        out = np.empty(self.n_dofs(l))
        check(self.lib.mgx_cube_seeded_vector(self.h, l, seed, out.ctypes.data_as(_lib.f64p)))
        return out


class LaplaceOperator:
    """multigrid::LaplaceOperator<3,p,number> of one level (laplace_operator.h:56-164)."""

    def __init__(self, ctx, desc=None, handle=None):
        self.ctx, self.lib = ctx, ctx.lib
        self.owned = handle is None
        if handle is None:
            h = C.c_void_p()
            check(self.lib.mgx_operator_create(ctx.h, C.byref(desc), C.byref(h)))
            self.h = h
        else:
            self.h = C.c_void_p(handle)
        self.number = self.lib.mgx_operator_number(self.h)

    @classmethod
    def from_cube(cls, ctx, cube, level, number=F64):
        return cls(ctx, cube.operator_desc(level, number))

    def m(self):
        return self.lib.mgx_operator_n_dofs(self.h)

    def set_profiled(self, on=True):
        check(self.lib.mgx_operator_set_profiled(self.h, int(on)))

    def initialize_dof_vector(self):
        return DeviceVector(self.ctx, self.m(), self.number)

    def vmult(self, dst, src):
        check(self.lib.mgx_vmult(self.h, dst.ptr, src.ptr))

    def vmult_with_cg_update(self, alpha, beta, r, q, p, x):
        """laplace_operator.h:638-719; returns the four sums {q.p, r.r, q.r, q.q}"""
        sums = np.zeros(4)
        check(self.lib.mgx_vmult_with_cg_update(self.h, alpha, beta, r.ptr, q.ptr, p.ptr, x.ptr, None,
                                                sums.ctypes.data_as(_lib.f64p)))
        return sums

    def vmult_residual(self, rhs, lhs, residual):
        check(self.lib.mgx_vmult_residual(self.h, rhs.ptr, lhs.ptr, residual.ptr))

    def compute_residual(self, dst, src=None, rhs_q=None):
        """LaplaceOperator::compute_residual (laplace_operator.h:804-845) on the device: dst = int f phi - A u_bc with the
        boundary values in the constrained entries of src and rhs_q = f JxW at the quadrature points (device vectors)"""
        check(self.lib.mgx_compute_residual(self.h, dst.ptr, src.ptr if src is not None else None,
                                            rhs_q.ptr if rhs_q is not None else None))

    def compute_diagonal(self):
        check(self.lib.mgx_compute_diagonal(self.h))

    def get_matrix_diagonal_inverse(self):
        p = C.c_void_p()
        check(self.lib.mgx_get_inverse_diagonal(self.h, C.byref(p)))
        return DeviceVector(self.ctx, self.m(), self.number, ptr=p)

    def clear(self):
        if self.owned and self.h:
            self.lib.mgx_operator_destroy(self.h)
            self.h = None


class Chebyshev:
    """dealii::PreconditionChebyshev<LaplaceOperator, Vector> (multigrid_solver.h:269-289)."""

    POLYNOMIAL = {"first_kind": 0, "fourth_kind": 1}

    def __init__(self, op, smoothing_range=20., degree=3, eig_cg_n_iterations=15, handle=None, polynomial=None):
        self.op, self.lib = op, op.lib
        self.owned = handle is None
        if handle is None:
            h = C.c_void_p()
            check(self.lib.mgx_smoother_create(op.h, smoothing_range, degree, eig_cg_n_iterations, C.byref(h)))
            self.h = h
        else:
            self.h = C.c_void_p(handle)
        if polynomial is not None:
            self.set_polynomial_type(polynomial)

    def set_polynomial_type(self, polynomial):
        """AdditionalData::polynomial_type: "first_kind" or "fourth_kind" (multigrid_solver.h:277, 951)"""
        check(self.lib.mgx_smoother_set_polynomial_type(self.h, self.POLYNOMIAL[polynomial]))

    def info(self):
        i = _lib.SmootherInfo()
        check(self.lib.mgx_smoother_get_info(self.h, C.byref(i)))
        return dict(lambda_min=i.lambda_min, lambda_max=i.lambda_max, theta=i.theta, delta=i.delta,
                    degree=i.degree, cg_its=i.cg_iterations)

    def vmult(self, x, b):
        check(self.lib.mgx_smoother_vmult(self.h, x.ptr, b.ptr))

    def step(self, x, b):
        check(self.lib.mgx_smoother_step(self.h, x.ptr, b.ptr))

    def clear(self):
        if self.owned and self.h:
            self.lib.mgx_smoother_destroy(self.h)
            self.h = None


class Transfer:
    """One level pair of dealii::MGTransferMatrixFree (multigrid_solver.h:209-222)."""

    def __init__(self, coarse, fine, children, prolong_1d):
        self.lib = coarse.lib
        self._children = np.ascontiguousarray(children, dtype=np.uint32)
        self._p1 = np.ascontiguousarray(prolong_1d, dtype=np.float64)
        d = _lib.TransferDesc(self._children.ctypes.data_as(_lib.u32p), self._p1.ctypes.data_as(_lib.f64p))
        h = C.c_void_p()
        check(self.lib.mgx_transfer_create(coarse.h, fine.h, C.byref(d), C.byref(h)))
        self.h = h

    def prolongate(self, fine, coarse, with_constraints=False):
        check(self.lib.mgx_prolongate(self.h, fine.ptr, coarse.ptr, 0, int(with_constraints)))

    def prolongate_and_add(self, fine, coarse, with_constraints=True):
        check(self.lib.mgx_prolongate(self.h, fine.ptr, coarse.ptr, 1, int(with_constraints)))

    def restrict_and_add(self, coarse, fine, with_constraints=True):
        check(self.lib.mgx_restrict_and_add(self.h, coarse.ptr, fine.ptr, int(with_constraints)))

    def clear(self):
        if self.h:
            self.lib.mgx_transfer_destroy(self.h)
            self.h = None


class MultigridSolver:
    """multigrid::MultigridSolver<3,p,Number,double> for poisson_cube (multigrid_solver.h:96-782).

    ctor arguments follow the reference: (dof_handler -> cube, degree_pre, degree_post, n_cycles);
    `vcycle_number` is the template parameter Number (program.cc:76: float; BASELINE: double)."""

    def __init__(self, ctx, cube, degree_pre=3, degree_post=3, n_cycles=1, vcycle_number=F64, comm=None,
                 polynomial="first_kind", agglomerate=True, device_rhs=False):
        """device_rhs: the right-hand sides are assembled on the GPU (mgx_solver_compute_rhs) instead of on the host.
        polynomial: Chebyshev polynomial type of the level smoothers: "first_kind" is what
        MultigridSolver<dim,p,Number,Number2> sets (multigrid_solver.h:277-278), "fourth_kind" what
        the Number == Number2 specialisation sets (:951-952)"""
        assert degree_pre == degree_post  # multigrid_solver.h:126
        self.ctx, self.cube, self.lib = ctx, cube, ctx.lib
        self.vnumber = vcycle_number
        self.comm = comm
        self.s = _lib.CubeSolver()
        if cube.size > 1:
            if comm is None:
                raise ValueError("a decomposed cube needs a Communicator")
        check(self.lib.mgx_cube_solver_create_opt(ctx.h, cube.h, vcycle_number, degree_pre, n_cycles, int(bool(device_rhs)),
                                                  C.byref(self.s)))
        self.n_levels = self.s.n_levels
        self.max_level = self.n_levels - 1
        self.h = C.c_void_p(self.s.solver)
        self.coarse = None
        if cube.size > 1 and (cube.box_desc is not None or cube.shell is not None) and \
                ((agglomerate and os.environ.get("MGX_AGGLOMERATE", "1") != "0") or cube.level_offset > 0):
            # the set-up is local; whether to use it is decided by all ranks together (a rank that
            # could not build its copy must not leave the others waiting in the allreduce)
            prepared = None
            try:
                prepared = self._agglomerate(degree_pre, n_cycles, vcycle_number)
            except Exception as e:  # noqa: BLE001
                import sys
                print("mgx MultigridSolver: coarse levels stay decomposed (%r)" % (e,), file=sys.stderr, flush=True)
            everyone = prepared is not None
            if comm is not None and hasattr(comm, "allreduce"):
                everyone = comm.allreduce([1.0 if prepared is not None else 0.0])[0] == cube.size
            if prepared is not None and everyone:
                coarse, level, mine, owned = prepared
                check(self.lib.mgx_solver_set_agglomeration(self.h, level, coarse.h, mine.ctypes.data_as(_lib.u32p),
                                                            owned.ctypes.data_as(C.POINTER(C.c_uint8)), mine.size))
                self.coarse, self.coarse_level = coarse, level
            elif prepared is not None:
                prepared[0].close()
        if polynomial != "first_kind":
            check(self.lib.mgx_solver_set_polynomial_type(self.h, Chebyshev.POLYNOMIAL[polynomial]))

    def _agglomerate(self, degree, n_cycles, vnumber):
        """Coarse levels of a decomposed hierarchy on every rank as a whole (mgx_solver_set_agglomeration):
        the levels whose global size is at most MGX_AGGLOMERATE_MAX_DOFS (default 3 000 000) -- there a level's
        work is microseconds and every exchange a latency.  (Up to 600 000 DoFs the copy is replayed as one HIP graph;
        the 2.1 M-DoF level of the 128^3 problem on top of it costs every rank 0.24 ms and a 17 MB allreduce instead
        of 0.45 ms of latency-bound exchanges: emulated rank at N = 8 2.71 -> 2.50 ms.)"""
        # (with the callback transport the allreduce is staged through the host: the seam stays where the vector is small)
        native = bool(getattr(self.comm, "native_ready", False)) or not hasattr(self.comm, "native_ready")
        cube, limit = self.cube, int(os.environ.get("MGX_AGGLOMERATE_MAX_DOFS", "3000000" if native else "600000"))
        level = -1
        # level i of a shell whose level-1 cells are dealt out to the ranks is level i + off of the whole mesh: the
        # coarse cells exist on the undecomposed copy only, which is therefore not optional there
        off = cube.level_offset
        for l in range(self.max_level):
            if cube.shell is not None:   # (cells x p^3 + the DoFs of two spherical boundary layers: an upper bound will do)
                size = cube.shell_desc["shell"] * 8 ** (l + off) * (cube.degree + 1) ** 3
            else:
                size = int((np.array(cube.cells_per_dim3(l)[1], dtype=np.int64) * cube.degree + 1).prod())
            if size <= limit:
                level = l
        if level < 0 and off > 0:
            level = 0   # (a mesh refined once: the rank's hierarchy is its finest level alone, the seam is that level)
        if level < 0:
            return None
        if cube.shell is not None:
            whole = Cube(cube.degree, n_refine=level + off, shell=cube.shell_desc["shell"], problem=cube.shell_desc["problem"])
        else:
            d = cube.box_desc
            whole = Cube(cube.degree, n_refine=level, box=d["box"], procs=(1, 1, 1), rank=0, numbering=d["numbering"],
                         origin=d["origin"], h0=d["h0"], geometry=d["geometry"], problem=d["problem"])
        ctx2 = Context(self.ctx.device)
        coarse = MultigridSolver(ctx2, whole, degree, degree, n_cycles, vnumber, device_rhs=True)  # its rhs is never used
        # local DoF -> DoF of the whole level through the run-independent id of a DoF
        gg = whole.dof_grid(level + off)
        order = np.argsort(gg)
        at = np.searchsorted(gg[order], cube.dof_grid(level))
        assert np.array_equal(gg[order][at], cube.dof_grid(level))
        mine = np.ascontiguousarray(order[at].astype(np.uint32))
        owned = np.ones(mine.size, dtype=np.uint8)
        owned[cube.not_owned(level)] = 0
        return coarse, level, mine, owned

    def matrix_dp(self, level):
        return LaplaceOperator(self.ctx, handle=self.s.matrix_dp[level])

    def matrix(self, level):
        return LaplaceOperator(self.ctx, handle=self.s.matrix[level])

    def smoother(self, level):
        p = C.c_void_p()
        check(self.lib.mgx_solver_get_smoother(self.h, level, C.byref(p)))
        return Chebyshev(self.matrix(level), handle=p.value)

    def solve(self, do_analyze=False, level_errors=False):
        """returns (reduction_rate, trace[n_levels,2] = residual norm start/end per level); with
        level_errors also errors[n_levels,2] = the L2 error of every level before / after its
        cycles, evaluated at the points of the solve where the reference prints them
        (multigrid_solver.h:420-424, 468-472)"""
        rate = C.c_double(1.0)
        trace = np.zeros(2 * self.n_levels)
        if not (do_analyze and level_errors):
            check(self.lib.mgx_solver_solve(self.h, int(do_analyze), C.byref(rate), trace.ctypes.data_as(_lib.f64p)))
            return rate.value, trace.reshape(-1, 2)
        errors = np.zeros((self.n_levels, 2))

        def hook(user, level, stage):
            errors[level, stage] = self.compute_l2_error(level)

        cb = _lib.LEVEL_HOOK_FN(hook)
        check(self.lib.mgx_solver_solve_hooked(self.h, 1, C.byref(rate), trace.ctypes.data_as(_lib.f64p), cb, None))
        return rate.value, trace.reshape(-1, 2), errors

    def solve_cg(self):
        its = C.c_uint()
        red = C.c_double()
        check(self.lib.mgx_solver_solve_cg(self.h, C.byref(its), C.byref(red)))
        return its.value, red.value

    def cg_history(self):
        """residual norms of the last solve_cg / solve_cg_fused: [0] at the start, [k] after iteration k"""
        out = np.zeros(1001)
        n = C.c_int()
        check(self.lib.mgx_solver_cg_history(self.h, out.ctypes.data_as(_lib.f64p), out.size, C.byref(n)))
        return out[:n.value].copy()

    def vmult(self, dst, src):
        check(self.lib.mgx_solver_vmult(self.h, dst.ptr, src.ptr))

    def vmult_with_residual_update(self, residual, update, factor):
        """multigrid_solver.h:516-619; returns {z.residual, z.(factor update)}"""
        out = np.zeros(2)
        check(self.lib.mgx_solver_vmult_with_residual_update(self.h, residual.ptr, update.ptr, factor,
                                                             out.ctypes.data_as(_lib.f64p)))
        return out

    def solve_cg_fused(self):
        its = C.c_uint()
        red = C.c_double()
        check(self.lib.mgx_solver_solve_cg_fused(self.h, C.byref(its), C.byref(red)))
        return its.value, red.value

    def do_matvec(self):
        check(self.lib.mgx_solver_do_matvec(self.h))

    def do_matvec_smoother(self):
        check(self.lib.mgx_solver_do_matvec_smoother(self.h))

    def get_solution(self, level=None, insert_bc=True):
        level = self.max_level if level is None else level
        p = C.c_void_p()
        check(self.lib.mgx_solver_get_solution(self.h, level, int(insert_bc), C.byref(p)))
        return DeviceVector(self.ctx, self.cube.n_dofs(level), F64, ptr=p)

    def get_vector(self, level, which):
        ids = dict(rhs=0, residual=1, defect=2, t=3, solution_update=4)
        p = C.c_void_p()
        check(self.lib.mgx_solver_get_vector(self.h, level, ids[which], C.byref(p)))
        number = F64 if ids[which] < 2 else self.vnumber
        return DeviceVector(self.ctx, self.cube.n_dofs(level), number, ptr=p)

    def compute_l2_error(self, level=None):
        level = self.max_level if level is None else level
        sol = self.get_solution(level, True).download()
        if self.cube.size == 1:
            return self.cube.l2_error(level, sol)
        e, v = self.comm.allreduce(self.cube.l2_error_parts(level, sol))
        return float(np.sqrt(e / v))

    def enable_timings(self, on=True):
        check(self.lib.mgx_solver_enable_timings(self.h, int(on)))

    def wall_times(self):
        t = np.zeros(6 * self.n_levels)
        check(self.lib.mgx_solver_get_timings(self.h, t.ctypes.data_as(_lib.f64p)))
        return t.reshape(-1, 6)

    def close(self):
        if getattr(self, "h", None):
            self.lib.mgx_cube_solver_destroy(C.byref(self.s))
            self.h = None
        if getattr(self, "coarse", None) is not None:
            c = self.coarse
            self.coarse = None
            c.close()
            c.cube.close()
            c.ctx.close()


# ---------------------------------------------------------------------------------------------
# DG path (include/mgx_dg.h)
DG_HERMITE, DG_GAUSS_LOBATTO, DG_GAUSS = 0, 1, 2


def dg_cheby_mesh(n_cell_steps):
    """cells per direction and cell Jacobian of matvec_dg_cheby/program.cc:55-77"""
    cells, jac = (C.c_int * 3)(), (C.c_double * 9)()
    check(_lib.load().mgx_dg_cheby_mesh(n_cell_steps, C.byref(cells), C.byref(jac)))
    return tuple(cells), np.array(jac).reshape(3, 3)


def dg_box_neighbours(cells, ordering="z"):
    """(neighbour table [n, 6], cell positions [n, 3]) of a box of cells, all outer faces Dirichlet"""
    c = (C.c_int * 3)(*cells)
    n = int(np.prod(cells))
    nb = np.empty((n, 6), dtype=np.int32)
    ijk = np.empty((n, 3), dtype=np.int32)
    i32p = C.POINTER(C.c_int32)
    check(_lib.load().mgx_dg_box_neighbours(C.byref(c), 1 if ordering == "z" else 0, nb.ctypes.data_as(i32p),
                                            ijk.ctypes.data_as(i32p)))
    return nb, ijk


def dg_partition(ijk, cells, procs, rank):
    """Ghost cells and exchange lists of a block decomposition, for the owned cells `ijk` ([n, 3] global
    positions, in the local cell order) of the rank whose block of the process grid they fill.
    Returns a dict: neighbours [n_owned, 6] (entries >= n_owned: ghost cells, -1: boundary), ijk,
    n_ghost, and exchange = [(rank, send_cells, recv_first, count)] in ascending rank order; the ghosts
    of one rank and the cells sent to it are in ascending global lexicographic order on both sides."""
    cells, procs = np.asarray(cells, dtype=np.int64), np.asarray(procs, dtype=np.int64)
    assert (cells % procs == 0).all(), "the process grid must divide the cells"
    blk = cells // procs
    ijk = np.asarray(ijk, dtype=np.int64)
    n_owned = len(ijk)
    gid = lambda p: p[..., 0] + cells[0] * (p[..., 1] + cells[1] * p[..., 2])   # noqa: E731
    owner = lambda p: (p[..., 0] // blk[0]) + procs[0] * ((p[..., 1] // blk[1]) + procs[1] * (p[..., 2] // blk[2]))  # noqa: E731
    assert (owner(ijk) == rank).all(), "cells outside the rank's block"
    pos = np.full(int(cells.prod()), -1, dtype=np.int64)   # global cell -> local index (owned, then ghosts)
    pos[gid(ijk)] = np.arange(n_owned)
    faces = []
    ghosts, sends = {}, {}
    for d in range(3):
        for s in (-1, 1):
            q = ijk.copy()
            q[:, d] += s
            ok = (q[:, d] >= 0) & (q[:, d] < cells[d])
            g = np.where(ok, gid(np.where(ok[:, None], q, ijk)), -1)
            own = np.where(ok, owner(np.where(ok[:, None], q, ijk)), rank)
            faces.append((ok, g))
            for rk in np.unique(own[ok & (own != rank)]):
                m = ok & (own == rk)
                ghosts.setdefault(int(rk), []).append(g[m])
                sends.setdefault(int(rk), []).append(gid(ijk[m]))
    exchange, first = [], n_owned
    for rk in sorted(ghosts):
        gl, sl = np.unique(np.concatenate(ghosts[rk])), np.unique(np.concatenate(sends[rk]))
        assert len(gl) == len(sl)
        pos[gl] = first + np.arange(len(gl))
        exchange.append((rk, pos[sl].astype(np.uint32), first, len(gl)))
        first += len(gl)
    nb = np.full((n_owned, 6), -1, dtype=np.int32)
    for f, (ok, g) in enumerate(faces):
        nb[ok, f] = pos[g[ok]]
    assert (nb[nb >= 0] < first).all()
    return dict(neighbours=nb, ijk=ijk.astype(np.int32), n_ghost=first - n_owned, exchange=exchange)


def dg_box_partition(cells, procs, rank, ordering="z"):
    """Block decomposition of a box of cells over a process grid (the DG counterpart of the Cube
    provider's decomposition; stands in for the p4est partition of the reference): the rank's block
    in z-order (or lexicographic order), then dg_partition()."""
    cells, procs = np.asarray(cells), np.asarray(procs)
    assert (cells % procs == 0).all(), "the process grid must divide the cells"
    blk = cells // procs
    r3 = np.array([rank % procs[0], (rank // procs[0]) % procs[1], rank // (procs[0] * procs[1])])
    loc = np.stack(np.meshgrid(np.arange(blk[0]), np.arange(blk[1]), np.arange(blk[2]), indexing="ij"), -1).reshape(-1, 3)
    if ordering == "z":
        def spread(v):
            out = np.zeros_like(v, dtype=np.int64)
            for bit in range(21):
                out |= ((v >> bit) & 1) << (3 * bit)
            return out
        key = spread(loc[:, 0]) | (spread(loc[:, 1]) << 1) | (spread(loc[:, 2]) << 2)
    else:
        key = loc[:, 0] + blk[0] * (loc[:, 1] + blk[1] * loc[:, 2])
    loc = loc[np.argsort(key, kind="stable")]
    # cells with a face on another rank last: the library then runs the leading (interior) cells
    # under the ghost exchange without an index list
    at_rank_face = np.zeros(len(loc), dtype=bool)
    for d in range(3):
        if r3[d] > 0:
            at_rank_face |= loc[:, d] == 0
        if r3[d] + 1 < procs[d]:
            at_rank_face |= loc[:, d] == blk[d] - 1
    loc = np.concatenate([loc[~at_rank_face], loc[at_rank_face]])
    return dg_partition(loc + r3 * blk, cells, procs, rank)


class DGLaplaceOperator:
    """multigrid::LaplaceOperatorCompactCombine<3,p,Number,type> with its JacobiTransformed
    preconditioner (common/laplace_operator_dg.h:350-2256) on an affine mesh."""

    def __init__(self, ctx, degree, basis, neighbours, jacobian, number=F32, n_ghost=0, exchange=None, plan_id=77):
        """n_ghost / exchange: decomposed mesh as dg_box_partition() describes it (the context then
        needs a Communicator)"""
        self.ctx, self.lib = ctx, ctx.lib
        self.degree, self.basis, self.number = degree, basis, number
        nb = np.ascontiguousarray(neighbours, dtype=np.int32).reshape(-1, 6)
        d = _lib.DGOperatorDesc()
        d.degree, d.basis, d.number, d.n_cells = degree, basis, number, nb.shape[0]
        d.neighbours = nb.ctypes.data_as(C.POINTER(C.c_int32))
        d.jacobian = (C.c_double * 9)(*np.asarray(jacobian, dtype=float).ravel())
        d.n_ghost_cells = n_ghost
        if n_ghost:
            k = len(exchange)
            ranks = (C.c_int * k)(*[e[0] for e in exchange])
            counts = (C.c_uint32 * k)(*[e[3] for e in exchange])
            first = (C.c_uint32 * k)(*[e[2] for e in exchange])
            lists = [np.ascontiguousarray(e[1], dtype=np.uint32) for e in exchange]
            ptrs = (_lib.u32p * k)(*[a.ctypes.data_as(_lib.u32p) for a in lists])
            ex = _lib.DGExchangeDesc(plan_id, k, C.cast(ranks, C.POINTER(C.c_int)), C.cast(counts, _lib.u32p),
                                     C.cast(ptrs, C.POINTER(_lib.u32p)), C.cast(first, _lib.u32p))
            d.exchange = C.pointer(ex)
        h = C.c_void_p()
        check(self.lib.mgx_dg_operator_create(ctx.h, C.byref(d), C.byref(h)))
        self.h = h

    def m(self):
        """owned DoFs of this rank"""
        return int(self.lib.mgx_dg_operator_n_dofs(self.h))

    def vector_size(self):
        """entries of a vector: owned cells, then ghost cells"""
        return int(self.lib.mgx_dg_operator_vector_size(self.h))

    def initialize_dof_vector(self, data=None):
        """data: values of the owned DoFs"""
        v = self.ctx.vector(self.vector_size(), self.number)
        if data is not None:
            a = np.zeros(self.vector_size(), dtype=_DT[self.number])
            a[:self.m()] = np.asarray(data).ravel()
            v.upload(a)
        return v

    def update_ghost_values(self, v):
        check(self.lib.mgx_dg_update_ghost_values(self.h, v.ptr))

    def vmult(self, dst, src):
        check(self.lib.mgx_dg_vmult(self.h, dst.ptr, src.ptr))

    def vmult_residual(self, dst, rhs, src):
        check(self.lib.mgx_dg_vmult_residual(self.h, dst.ptr, rhs.ptr, src.ptr))

    def jacobi_vmult(self, dst, src):
        check(self.lib.mgx_dg_jacobi_vmult(self.h, dst.ptr, src.ptr))

    def vmult_with_chebyshev_update(self, rhs, iteration_index, factor1, factor2, solution, solution_old):
        """as the reference (laplace_operator_dg.h:910-955): on return `solution` holds the new
        iterate and `solution_old` the previous one (the two vectors trade their storage)"""
        check(self.lib.mgx_dg_vmult_with_chebyshev_update(self.h, rhs.ptr, iteration_index, factor1, factor2,
                                                          solution.ptr, solution_old.ptr))
        if iteration_index > 0:
            solution.ptr, solution_old.ptr = solution_old.ptr, solution.ptr
            solution.owned, solution_old.owned = solution_old.owned, solution.owned

    def vmult_with_cg_update(self, alpha, beta, r, q, p, x):
        """one merged CG iteration (laplace_operator_dg.h:863-908): x += alpha p, p = beta p + q (alpha == 0: p = q),
        q = A p; returns (q.p, r.r, q.r, q.q) summed over the ranks"""
        sums = np.empty(4)
        check(self.lib.mgx_dg_vmult_with_cg_update(self.h, alpha, beta, r.ptr, q.ptr, p.ptr, x.ptr,
                                                   sums.ctypes.data_as(_lib.f64p)))
        return sums

    def basis_1d(self):
        """(shape values [q, i] in the Gauss points, Gauss points, Gauss weights) on [0, 1]"""
        n = self.degree + 1
        S, x, w = np.empty((n, n)), np.empty(n), np.empty(n)
        check(self.lib.mgx_dg_operator_basis(self.h, S.ctypes.data_as(_lib.f64p), x.ctypes.data_as(_lib.f64p),
                                             w.ctypes.data_as(_lib.f64p)))
        return S, x, w

    def info(self):
        hd = C.c_double()
        pen = (C.c_double * 3)()
        ev = (C.c_double * 10)()
        check(self.lib.mgx_dg_operator_info(self.h, C.byref(hd), pen, ev))
        return dict(hermite_derivative_on_face=hd.value, penalty=np.array(pen),
                    eigenvalues_1d=np.array(ev)[:self.degree + 1])

    def clear(self):
        if getattr(self, "h", None):
            self.lib.mgx_dg_operator_destroy(self.h)
            self.h = None


class DGMultigridSolver:
    """multigrid::MultigridSolverDG<3,p,Number,double> (common/multigrid_solver_dg.h:55-747) on the
    Cartesian meshes of the Cube provider: a DG level (operator, block-Jacobi Chebyshev smoother) on
    top of the FE_Q(p) hierarchy of the same mesh, V-cycle in `vcycle_number`, outer CG in fp64.
    The DG cells are the cells of the cube's finest level in the provider's order."""

    def __init__(self, ctx, cube, basis=DG_HERMITE, degree_pre=3, vcycle_number=F32, comm=None):
        """comm: Communicator of a decomposed cube (box form): the DG cells get ghost cells, the FE_Q
        hierarchy is the decomposed one (not agglomerated: its smoothers are re-configured)"""
        self.ctx, self.cube, self.lib = ctx, cube, ctx.lib
        l = cube.max_level
        local, whole = (np.array(v, dtype=np.int64) for v in cube.cells_per_dim3(l))
        procs = whole // local
        r3 = np.array([cube.rank % procs[0], (cube.rank // procs[0]) % procs[1], cube.rank // (procs[0] * procs[1])])
        ijk = cube.cell_coords(l).astype(np.int64) + r3 * local      # global positions, provider's cell order
        part = dg_partition(ijk, whole, procs, cube.rank)
        self.neighbours, self.cell_ijk = part["neighbours"], part["ijk"]
        gid = (ijk[:, 0] + whole[0] * (ijk[:, 1] + whole[1] * ijk[:, 2])).astype(np.uint32)
        jac = np.eye(3) * cube.cell_size(l)
        self.cfe = MultigridSolver(ctx, cube, degree_pre, degree_pre, 1, vcycle_number, comm=comm, agglomerate=False)
        ng, ex = part["n_ghost"], part["exchange"]
        self.matrix_dg = DGLaplaceOperator(ctx, cube.degree, basis, part["neighbours"], jac, vcycle_number, ng, ex, plan_id=1001)
        self.matrix_dg_dp = DGLaplaceOperator(ctx, cube.degree, basis, part["neighbours"], jac, F64, ng, ex, plan_id=1002)
        d = _lib.DGSolverDesc(self.matrix_dg.h, self.matrix_dg_dp.h, self.cfe.h, degree_pre, gid.ctypes.data_as(_lib.u32p))
        h = C.c_void_p()
        check(self.lib.mgx_dg_solver_create(ctx.h, C.byref(d), C.byref(h)))
        self.h = h
        self.vnumber = vcycle_number
        self.cell_gid = gid

    def initialize_dof_vector(self, data=None):
        """fp64 vector of the solver interface (owned DoFs, then the ghost cells of a decomposed mesh)"""
        return self.matrix_dg_dp.initialize_dof_vector(data)

    def m(self):
        return self.matrix_dg.m()

    def smoother_info(self):
        i = _lib.SmootherInfo()
        check(self.lib.mgx_dg_solver_smoother_info(self.h, C.byref(i)))
        return dict(lambda_min=i.lambda_min, lambda_max=i.lambda_max, theta=i.theta, delta=i.delta,
                    degree=i.degree, cg_its=i.cg_iterations)

    def vmult(self, dst, src):
        """one DG V-cycle (multigrid_solver_dg.h:429-440); fp64 vectors"""
        check(self.lib.mgx_dg_solver_vmult(self.h, dst.ptr, src.ptr))

    def solve_cg(self, rhs, solution, tolerance=1e-9):
        """(iterations, reduction rate per iteration) of the V-cycle-preconditioned CG (:410-424)"""
        its, red = C.c_uint(), C.c_double()
        check(self.lib.mgx_dg_solver_solve_cg(self.h, tolerance, rhs.ptr, solution.ptr, C.byref(its), C.byref(red)))
        return its.value, red.value

    def restrict_to_cg(self, cg_dst, dg_src):
        check(self.lib.mgx_dg_restrict_to_cg(self.h, cg_dst.ptr, dg_src.ptr))

    def prolongate_add_cg_to_dg(self, dg_dst, cg_src):
        check(self.lib.mgx_dg_prolongate_add_cg_to_dg(self.h, dg_dst.ptr, cg_src.ptr))

    def vmult_residual_and_restrict_to_cg(self, cg_dst, rhs, lhs):
        """cg = P^T (rhs - A lhs) in one kernel (laplace_operator_dg.h:852-861), V-cycle number type"""
        check(self.lib.mgx_dg_vmult_residual_and_restrict_to_cg(self.h, cg_dst.ptr, rhs.ptr, lhs.ptr))

    def close(self):
        if getattr(self, "h", None):
            self.lib.mgx_dg_solver_destroy(self.h)
            self.h = None
            self.matrix_dg.clear()
            self.matrix_dg_dp.clear()
            self.cfe.close()
