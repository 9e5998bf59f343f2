"""`mpi4py.MPI` stand-in: P ranks simulated as P threads (TEST INFRASTRUCTURE).

API surface = what the reference calls on its hot path (reference files, all
relative to /root/reference):
  pyDNMFk/dist_comm.py:16-56   Get_rank/Get_size/Create_cart/Get_coords/Sub/Free
  pyDNMFk/dist_nmf.py:114,163,169,195,202  allreduce/allgather/Reduce_scatter/barrier
  pyDNMFk/pyDNMF.py:121,129,189,217        bcast/allreduce
  pyDNMFk/utils.py:78-93                   allreduce of ints, barrier

Semantics notes
  * Reductions are rank-ordered sequential sums ((r0+r1)+r2)+...  Real MPI
    picks an implementation-defined order, so parity against these fixtures is
    a tolerance, never bit-equality (SURVEY.md section 2.4 numerical note).
  * Object collectives return private copies per rank (mpi4py pickles), so
    ranks never alias each other's arrays.
  * Reduce_scatter scatters by each rank's actual recvbuf size (the sensible
    extension of MPI_Reduce_scatter to ragged blocks).

Usage: `run_ranks(P, fn)` runs fn(rank) on P threads; inside, `COMM_WORLD`
resolves to the calling thread's view of the world communicator.
"""
import copy
import threading
import time

import numpy as np

SUM = "SUM"

_tls = threading.local()


def Wtime():
    return time.time()


class _Group:
    """State shared by all member threads of one communicator."""

    def __init__(self, members):
        self.members = list(members)          # world ranks, in communicator-rank order
        self.size = len(self.members)
        self.barrier = threading.Barrier(self.size)
        self.slots = [None] * self.size
        self.lock = threading.Lock()
        self.children = {}                    # (call_idx, key) -> _Group


class Comm:
    """Per-thread handle on a communicator."""

    def __init__(self, group, rank, dims=None):
        self._g = group
        self.rank = rank
        self.size = group.size
        self._dims = dims
        self._ncalls = 0                      # per-thread count of collective constructors

    # --- introspection ---
    def Get_rank(self):
        return self.rank

    def Get_size(self):
        return self.size

    # --- plumbing ---
    def _exchange(self, obj):
        g = self._g
        g.slots[self.rank] = obj
        g.barrier.wait()
        vals = list(g.slots)
        g.barrier.wait()
        return vals

    def _child(self, key, members_fn):
        g = self._g
        idx = self._ncalls
        self._ncalls += 1
        with g.lock:
            ck = (idx, key)
            if ck not in g.children:
                g.children[ck] = _Group(members_fn())
            return g.children[ck]

    # --- collectives ---
    def barrier(self):
        self._g.barrier.wait()

    Barrier = barrier

    def allreduce(self, obj, op=SUM):
        vals = self._exchange(obj)
        acc = copy.deepcopy(vals[0])
        for v in vals[1:]:
            acc = acc + v
        return acc

    def allgather(self, obj):
        return [copy.deepcopy(v) for v in self._exchange(obj)]

    def bcast(self, obj, root=0):
        return copy.deepcopy(self._exchange(obj)[root])

    def Bcast(self, buf, root=0):
        vals = self._exchange(buf)
        if self.rank != root:
            buf[...] = vals[root]

    def scatter(self, objs, root=0):
        return copy.deepcopy(self._exchange(objs)[root][self.rank])

    def Reduce_scatter(self, sendbuf, recvbuf, recvcounts=None, op=SUM):
        vals = self._exchange((np.ascontiguousarray(sendbuf), recvbuf.size))
        acc = vals[0][0].reshape(-1).copy()
        for v, _ in vals[1:]:
            acc = acc + v.reshape(-1)
        counts = [c for _, c in vals]
        off = sum(counts[: self.rank])
        recvbuf.reshape(-1)[...] = acc[off: off + counts[self.rank]]

    # --- topology ---
    def Create_cart(self, dims, periods=None, reorder=False):
        assert int(np.prod(dims)) == self.size
        grp = self._child(("cart", tuple(dims)), lambda: self._g.members)
        return Comm(grp, self.rank, dims=tuple(dims))

    def Get_coords(self, rank):
        return [int(c) for c in np.unravel_index(rank, self._dims)]

    def Sub(self, remain_dims):
        coords = self.Get_coords(self.rank)
        fixed = tuple(c for c, keep in zip(coords, remain_dims) if not keep)
        sub_dims = tuple(d for d, keep in zip(self._dims, remain_dims) if keep)

        def members():
            out = []
            for r in range(self.size):
                cr = self.Get_coords(r)
                if tuple(c for c, keep in zip(cr, remain_dims) if not keep) == fixed:
                    out.append(r)
            return out

        mem = members()
        grp = self._child(("sub", tuple(remain_dims), fixed), lambda: mem)
        return Comm(grp, mem.index(self.rank), dims=sub_dims)

    def Free(self):
        pass


class _WorldProxy:
    """`MPI.COMM_WORLD`: forwards to the calling thread's world handle."""

    def __getattr__(self, name):
        return getattr(_tls.world, name)


COMM_WORLD = _WorldProxy()


def run_ranks(nranks, fn):
    """Run fn(rank) on `nranks` threads; returns the per-rank results."""
    world = _Group(range(nranks))
    results = [None] * nranks
    errors = [None] * nranks

    def body(r):
        _tls.world = Comm(world, r)
        try:
            results[r] = fn(r)
        except BaseException as e:  # noqa: BLE001
            errors[r] = e
            world.barrier.abort()

    threads = [threading.Thread(target=body, args=(r,)) for r in range(nranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in errors:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    for e in errors:
        if e is not None:
            raise e
    return results
