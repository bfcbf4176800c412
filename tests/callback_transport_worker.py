"""One rank, backend nccl (= RCCL): the callback transport of multigrid_amd.Communicator on device memory,
with buffers the alloc callback never saw -- what the DG ghost exchange hands it (packed sends from the
library's own allocations, receives straight into the ghost part of a vector).  The rank sends to itself."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    import multigrid_amd as mg
    ctx = mg.Context(0)
    comm = mg.Communicator(ctx, dist, device_transport=True, native=False)
    n = 1000
    for number, dt in ((mg.F64, np.float64), (mg.F32, np.float32)):
        for trial in range(2):  # second round: another pair of vectors under the same plan id
            a = ctx.vector(n, number, np.arange(n).astype(dt) + trial)
            b = ctx.vector(n, number)
            ctx.sync()
            ranks, counts = (C.c_int * 1)(0), (C.c_uint32 * 1)(n)
            send, recv = (C.c_void_p * 1)(a.ptr), (C.c_void_p * 1)(b.ptr)
            assert comm._exchange(None, 7, number, 1, ranks, counts, send, recv) == 0
            assert np.array_equal(b.download(), np.arange(n).astype(dt) + trial), (number, trial)
    # the wrapper is a view, not a copy
    v = ctx.vector(16, mg.F64, np.zeros(16))
    ctx.sync()
    t = comm._device_tensor(v.ptr.value, 128).view(torch.float64)
    t += 3.0
    torch.cuda.synchronize()
    assert np.array_equal(v.download(), np.full(16, 3.0))
    print("callback transport ok", flush=True)
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
