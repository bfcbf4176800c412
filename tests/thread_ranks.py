"""Ranks as threads of one process: an object with the part of the torch.distributed interface the Communicator
and tests/dist_worker.py use (point-to-point batches, all_reduce, broadcast, barrier), moving host tensors through
queues.  A one-GPU test box admits at most six GPU processes; the 2x2x2 layout of an 8-GPU run -- DoFs shared by four
and eight ranks, seven neighbours per rank -- runs here as eight threads on the one GPU (ctypes releases the GIL
around every library call; the exchange callback blocks in a queue, not in the interpreter)."""
import queue
import threading


class World:
    def __init__(self, size):
        self.size = size
        self.box = {(a, b): queue.Queue() for a in range(size) for b in range(size)}
        self.bar = threading.Barrier(size)
        self.slots = [None] * size
        self.failed = threading.Event()

    def barrier(self):
        self.bar.wait(timeout=600)


class _Req:
    def wait(self):
        return None


class _Op:
    SUM, MIN, MAX = "sum", "min", "max"


class ThreadDist:
    ReduceOp = _Op

    def __init__(self, world, rank):
        self.w, self.rank = world, rank

    def get_rank(self):
        return self.rank

    def get_world_size(self):
        return self.w.size

    def get_backend(self):
        return "gloo"

    # the two markers and the record torch.distributed.P2POp builds from them
    @staticmethod
    def isend(*a):
        raise NotImplementedError

    @staticmethod
    def irecv(*a):
        raise NotImplementedError

    class P2POp:
        def __init__(self, op, tensor, peer):
            self.send, self.tensor, self.peer = op is ThreadDist.isend, tensor, peer

    def batch_isend_irecv(self, ops):
        for o in ops:  # all sends first: queues are unbounded, nobody waits for a receiver
            if o.send:
                self.w.box[(self.rank, o.peer)].put(o.tensor.clone())
        for o in ops:
            if not o.send:
                o.tensor.copy_(self.w.box[(o.peer, self.rank)].get(timeout=600))
        return [_Req() for _ in ops]

    def all_reduce(self, t, op="sum"):
        w = self.w
        w.slots[self.rank] = t.clone()
        w.barrier()
        acc = w.slots[0].clone()
        for r in range(1, w.size):  # the same order on every rank
            if op == "sum":
                acc += w.slots[r]
            elif op == "min":
                acc = acc.minimum(w.slots[r])
            else:
                acc = acc.maximum(w.slots[r])
        w.barrier()
        t.copy_(acc)

    def broadcast(self, t, src):
        w = self.w
        if self.rank == src:
            w.slots[src] = t.clone()
        w.barrier()
        t.copy_(w.slots[src])
        w.barrier()

    def barrier(self):
        self.w.barrier()


def run_ranks(size, fn):
    """fn(dist, rank) on `size` threads; returns the list of results, re-raises the first failure"""
    world = World(size)
    out, err = [None] * size, [None] * size

    def body(r):
        try:
            out[r] = fn(ThreadDist(world, r), r)
        except BaseException as e:  # noqa: BLE001
            err[r] = e
            world.failed.set()
            world.bar.abort()  # the other ranks must not wait for this one

    threads = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(size)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=900)
    first = next((e for e in err if e is not None and not isinstance(e, threading.BrokenBarrierError)), None) or \
        next((e for e in err if e is not None), None)
    if first is not None:
        raise first
    assert all(not t.is_alive() for t in threads), "a rank is still running"
    return out
