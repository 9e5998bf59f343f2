"""Cartesian p_r x p_c process grid over torch.distributed (RCCL on GPUs, gloo in CPU tests).

Mirrors reference pyDNMFk/dist_comm.py:16-56 (`MPI_comm`) and the slice of the mpi4py
communicator API the MU path uses.  One process per GPU; rank r sits at grid
coordinates (i, j) = divmod(r, p_c) (row-major, `Create_cart(..., reorder=False)`,
dist_comm.py:22).  Naming follows the reference:
  cart_1d_row()    = Sub([True, False])  -> the p_r ranks that share grid column j
  cart_1d_column() = Sub([False, True])  -> the p_c ranks that share grid row i
The reference's barrier after every collective (dist_nmf.py:115,139,164,...) is dropped:
collectives here are stream-ordered.
"""
import numpy as np
import torch
import torch.distributed as dist


def _dist_on():
    return dist.is_available() and dist.is_initialized()


class TorchComm:
    """Communicator handle: a torch.distributed process group with an mpi4py-shaped surface."""

    # A communicator of one rank needs no exchange and every collective below returns at once.  Set to True (tests do, on
    # a one-GPU box) a single-rank communicator still issues the real torch.distributed call, so that the RCCL code path
    # -- device buffers, sub-groups, stream ordering against the HIP kernels -- runs without a second GPU.
    always_collective = False

    def _solo(self):
        return self.size == 1 and not (TorchComm.always_collective and _dist_on())

    def __init__(self, group=None, ranks=None):
        self.group = group
        if _dist_on():
            self.world_rank = dist.get_rank()
            self.ranks = list(ranks) if ranks is not None else list(range(dist.get_world_size()))
            self.rank = self.ranks.index(self.world_rank)
            self.size = len(self.ranks)
            self.backend = dist.get_backend()
            self.device = (torch.device("cuda", torch.cuda.current_device())
                           if self.backend == "nccl" else torch.device("cpu"))
        else:
            self.world_rank, self.ranks, self.rank, self.size = 0, [0], 0, 1
            self.device = torch.device("cpu")
            self.backend = None

    # ---- mpi4py-shaped surface
    def Get_rank(self):
        return self.rank

    def Get_size(self):
        return self.size

    def barrier(self):
        if not self._solo():
            if self.backend == "nccl":
                dist.barrier(group=self.group, device_ids=[self.device.index])
            else:
                dist.barrier(group=self.group)

    Barrier = barrier

    def Free(self):
        pass

    def _to_tensor(self, x):
        if isinstance(x, torch.Tensor):
            return x.clone(), "t"
        if isinstance(x, np.ndarray):
            return torch.from_numpy(np.ascontiguousarray(x)).to(self.device), "n"
        if isinstance(x, (int, np.integer)):
            return torch.tensor([int(x)], dtype=torch.int64, device=self.device), "i"
        return torch.tensor([float(x)], dtype=torch.float64, device=self.device), "f"

    @staticmethod
    def _from_tensor(t, kind, like):
        if kind == "t":
            return t
        if kind == "n":
            return t.cpu().numpy().astype(like.dtype, copy=False)
        return int(t.item()) if kind == "i" else float(t.item())

    def allreduce(self, x, op=None):
        """SUM allreduce returning a new object (mpi4py lowercase semantics)."""
        if self._solo():
            return x
        t, kind = self._to_tensor(x)
        self.allreduce_(t)
        return self._from_tensor(t, kind, x)

    def bcast(self, x, root=0):
        if self._solo():
            return x
        if isinstance(x, torch.Tensor):
            c = self._wire(x).clone()
            dist.broadcast(c, src=self.ranks[root], group=self.group)
            return c.to(x.device)
        box = [x]
        dist.broadcast_object_list(box, src=self.ranks[root], group=self.group)
        return box[0]

    def allgather(self, x):
        if self._solo():
            return [x]
        if isinstance(x, torch.Tensor):
            shapes = [None] * self.size
            dist.all_gather_object(shapes, tuple(x.shape), group=self.group)
            return self.allgather_blocks(x, shapes)
        out = [None] * self.size
        dist.all_gather_object(out, x, group=self.group)
        return out

    # ---- device collectives used by the update choreography
    def _staged(self, t):
        """A tensor the transport cannot take where it lives: gloo (tests / debugging) moves bytes through host memory,
        RCCL only works on device buffers (the NMFk driver hands CPU tensors to the clustering when the data came in as
        numpy).  Such tensors make the trip through the other memory; the product path (device tensors over RCCL) never
        stages."""
        return (self.backend == "gloo" and t.is_cuda) or (self.backend == "nccl" and not t.is_cuda)

    def _wire(self, t):
        """`t` where the transport wants it (a copy iff staged)."""
        if not self._staged(t):
            return t
        return t.cpu() if self.backend == "gloo" else t.to(self.device)

    def allreduce_(self, t):
        if not self._solo():
            if self._staged(t):
                c = self._wire(t)
                dist.all_reduce(c, group=self.group)
                t.copy_(c)
            else:
                dist.all_reduce(t, group=self.group)
        return t

    def allreduce_begin(self, t):
        """Start a SUM allreduce of `t` in place and return a handle whose wait() orders the result before whatever
        the caller enqueues next.  Over RCCL the collective runs on the communicator's own stream (it starts when the
        kernels enqueued so far on the current stream are done), so kernels launched between begin and wait overlap it."""
        if self._solo():
            return _Done()
        if self._staged(t):
            self.allreduce_(t)                  # staged transports (tests) exchange synchronously
            return _Done()
        return dist.all_reduce(t, group=self.group, async_op=True)

    def allgather_blocks(self, x, shapes):
        """All-gather row-major blocks whose per-rank shapes are known (ragged allowed: padded to the largest)."""
        if self._solo():
            return [x]
        numels = [int(np.prod(s)) for s in shapes]
        mx = max(numels)
        send = x.reshape(-1)
        if send.numel() < mx:
            send = torch.cat([send, send.new_zeros(mx - send.numel())])
        c = self._wire(send.contiguous())
        rc = c.new_empty(self.size * mx)
        dist.all_gather_into_tensor(rc, c, group=self.group)
        recv = rc.to(send.device)
        return [recv[q * mx: q * mx + numels[q]].view(*shapes[q]) for q in range(self.size)]

    def allgather_blocks_begin(self, x, shapes):
        """Start `allgather_blocks(x, shapes)` and return a handle whose wait() gives the list of blocks.  Over RCCL (and gloo on
        host tensors) the collective runs asynchronously: it starts when the kernels enqueued so far have produced `x`, and
        whatever the caller launches before wait() overlaps it.  `x` must not change until wait()."""
        if self._solo() or self._staged(x.reshape(-1)):
            return _Ready(self.allgather_blocks(x, shapes))
        numels = [int(np.prod(s)) for s in shapes]
        mx = max(numels)
        send = x.reshape(-1)
        if send.numel() < mx:
            send = torch.cat([send, send.new_zeros(mx - send.numel())])
        send = send.contiguous()
        rc = send.new_empty(self.size * mx)
        work = dist.all_gather_into_tensor(rc, send, group=self.group, async_op=True)
        return _Pending(work, lambda: [rc[q * mx: q * mx + numels[q]].view(*shapes[q]) for q in range(self.size)], keep=(send, rc))

    def reduce_scatter_rows(self, full, counts):
        """SUM reduce-scatter of a (sum(counts) x c) row-major buffer by row blocks (MPI Reduce_scatter, dist_nmf.py:169,202).
        Over RCCL the wire carries what the reference's does -- one block per member: equal blocks go straight into
        reduce_scatter_tensor; RAGGED blocks (a dimension that does not divide: real data) are first laid out at the pitch
        of the largest one (p block copies of this rank's buffer, no zero fill: the padding rows are reduced into rows nobody
        reads), instead of an allreduce of the whole buffer, which moves p times the bytes.  gloo (tests) has no
        reduce-scatter: allreduce + slice, same sums."""
        if self._solo():
            return full
        c = full.shape[1]
        if self.backend != "gloo" and not self._staged(full):
            if len(set(counts)) == 1:
                out = full.new_empty(counts[0], c)
                dist.reduce_scatter_tensor(out, full.contiguous(), group=self.group)
                return out
            mx = max(counts)
            padded = full.new_empty(self.size, mx, c)
            off = 0
            for q, cnt in enumerate(counts):
                padded[q, :cnt].copy_(full[off: off + cnt])
                off += cnt
            out = full.new_empty(mx, c)
            dist.reduce_scatter_tensor(out, padded.view(self.size * mx, c), group=self.group)
            return out[: counts[self.rank]]
        t = full.clone()
        self.allreduce_(t)
        off = sum(counts[: self.rank])
        return t[off: off + counts[self.rank]].contiguous()

    def Reduce_scatter(self, sendbuf, recvbuf, op=None):
        counts = [None] * self.size
        if self.size == 1:
            recvbuf.copy_(sendbuf.view_as(recvbuf))
            return
        dist.all_gather_object(counts, int(recvbuf.shape[0]), group=self.group)
        recvbuf.copy_(self.reduce_scatter_rows(sendbuf, counts))


class _Done:
    """Handle of an exchange that has already happened."""

    def wait(self):
        return True


class _Ready:
    """Handle of a gather that has already happened: wait() hands its result over."""

    def __init__(self, value):
        self.value = value

    def wait(self):
        return self.value


class _Pending:
    """Handle of an asynchronous gather: wait() orders it before what the caller enqueues next and builds the result."""

    def __init__(self, work, result, keep=()):
        self.work, self.result, self.keep = work, result, keep

    def wait(self):
        self.work.wait()
        return self.result()


class NullExchange:
    """A communicator whose exchanges return at once (rank / size of the wrapped one).  MEASUREMENT ONLY: bench.py times
    the step with it to separate a rank's compute from the exchange; on more than one rank the results are wrong by
    construction (every rank keeps its own partial sums)."""

    def __init__(self, comm):
        self.rank, self.size, self.ranks = comm.rank, comm.size, comm.ranks
        self.backend, self.device, self.group = comm.backend, comm.device, comm.group

    def Get_rank(self):
        return self.rank

    def Get_size(self):
        return self.size

    def allreduce_(self, t):
        return t

    def allreduce_begin(self, t):
        return _Done()

    def allreduce(self, x, op=None):
        return x

    def allgather_blocks_begin(self, x, shapes):
        return _Ready(self.allgather_blocks(x, shapes))

    def allgather_blocks(self, x, shapes):
        """every member's block = a copy of this rank's (right sizes for the kernels that follow, no exchange)"""
        if self.size == 1:
            return [x]
        mx = max(int(np.prod(sh)) for sh in shapes)
        send = x.reshape(-1)
        rc = send.new_empty(self.size * mx)
        for q in range(self.size):
            rc[q * mx: q * mx + send.numel()].copy_(send)
        return [rc[q * mx: q * mx + int(np.prod(shapes[q]))].view(*shapes[q]) for q in range(self.size)]

    def reduce_scatter_rows(self, full, counts):
        if self.size == 1:
            return full
        off = sum(counts[: self.rank])
        return full[off: off + counts[self.rank]].contiguous()

    def bcast(self, x, root=0):
        return x

    def barrier(self):
        pass


class EmulatedGroup:
    """MEASUREMENT ONLY (bench.py --config 4 --emulate-ranks R): one member's view of a `size`-member sub-communicator on a
    box with one GPU.  Every collective issues the REAL torch.distributed call on the one-rank group `comm` (the call's fixed
    costs -- launch, stream ordering against the kernels -- are real; there is no wire) and hands back buffers of the sizes a
    real group would: the other members' allgather blocks are copies of this rank's, a reduce-scatter returns this member's
    block.  Results are not a factorisation of anything."""

    def __init__(self, comm, size, rank=0):
        self.inner, self.size, self.rank = comm, int(size), int(rank)
        self.ranks, self.backend, self.device, self.group = list(range(self.size)), comm.backend, comm.device, comm.group

    def Get_rank(self):
        return self.rank

    def Get_size(self):
        return self.size

    def allreduce_(self, t):
        return self.inner.allreduce_(t)

    def allreduce_begin(self, t):
        return self.inner.allreduce_begin(t)

    def allreduce(self, x, op=None):
        return self.inner.allreduce(x)

    def bcast(self, x, root=0):
        return self.inner.bcast(x, root=0)

    def barrier(self):
        self.inner.barrier()

    def allgather_blocks(self, x, shapes):
        mx = max(int(np.prod(sh)) for sh in shapes)
        send = x.reshape(-1)
        if send.numel() < mx:
            send = torch.cat([send, send.new_zeros(mx - send.numel())])
        rc = send.new_empty(self.size * mx)
        dist.all_gather_into_tensor(rc[:mx], send.contiguous(), group=self.inner.group)      # this member's block, really gathered
        for q in range(1, self.size):
            rc[q * mx: (q + 1) * mx].copy_(rc[:mx])
        return [rc[q * mx: q * mx + int(np.prod(shapes[q]))].view(*shapes[q]) for q in range(self.size)]

    def allgather_blocks_begin(self, x, shapes):
        """the asynchronous form (params.overlap_2d): the real call is issued with async_op, the copies follow in wait()"""
        mx = max(int(np.prod(sh)) for sh in shapes)
        send = x.reshape(-1)
        if send.numel() < mx:
            send = torch.cat([send, send.new_zeros(mx - send.numel())])
        send = send.contiguous()
        rc = send.new_empty(self.size * mx)
        work = dist.all_gather_into_tensor(rc[:mx], send, group=self.inner.group, async_op=True)

        def result():
            for q in range(1, self.size):
                rc[q * mx: (q + 1) * mx].copy_(rc[:mx])
            return [rc[q * mx: q * mx + int(np.prod(shapes[q]))].view(*shapes[q]) for q in range(self.size)]
        return _Pending(work, result, keep=(send, rc))

    def reduce_scatter_rows(self, full, counts):
        c = full.shape[1]
        out = full.new_empty(counts[self.rank], c)
        off = sum(counts[: self.rank])
        dist.reduce_scatter_tensor(out, full[off: off + counts[self.rank]].contiguous(), group=self.inner.group)
        return out


def COMM_WORLD():
    """The world communicator (every rank of the torch.distributed job, or a single process)."""
    return TorchComm(None)


class MPI_comm:
    """Reference dist_comm.py:16-56.  `comm` may be None (world) or a TorchComm."""

    def __init__(self, comm, p_r, p_c):
        self.comm = comm if comm is not None else COMM_WORLD()
        self.rank = self.comm.Get_rank()
        self.size = self.comm.Get_size()
        self.p_r, self.p_c = int(p_r), int(p_c)
        if self.p_r * self.p_c != self.size:
            raise ValueError("grid %dx%d needs %d ranks, communicator has %d" % (p_r, p_c, p_r * p_c, self.size))
        self.coord2d = list(divmod(self.rank, self.p_c))       # Create_cart(reorder=False).Get_coords
        self._row = self._col = None

    def _make(self, groups):
        mine = None
        for ranks in groups:                                   # new_group is collective over the world
            world = [self.comm.ranks[r] for r in ranks]
            whole = len(world) == self.size                    # a 1D grid's long axis: reuse the parent group
            g = dist.new_group(world) if (self.size > 1 and len(world) > 1 and not whole) else None
            if whole:
                g = self.comm.group
            if self.rank in ranks:
                if self.size == 1:
                    mine = TorchComm(None)
                elif len(world) == 1:
                    mine = _SelfComm(world[0])
                else:
                    mine = TorchComm(g, world)
        return mine

    def cart_1d_row(self):
        """Ranks with the same grid column j (size p_r, ordered by i) -- dist_comm.py:25-37."""
        if self._row is None:
            self._row = self._make([[i * self.p_c + j for i in range(self.p_r)] for j in range(self.p_c)])
        self.cartesian1d_row = self._row
        return self._row

    def cart_1d_column(self):
        """Ranks with the same grid row i (size p_c, ordered by j) -- dist_comm.py:39-51."""
        if self._col is None:
            self._col = self._make([[i * self.p_c + j for j in range(self.p_c)] for i in range(self.p_r)])
        self.cartesian1d_column = self._col
        return self._col

    def Free(self):
        """dist_comm.py:53-56 (the reference re-creates and frees the sub-communicators; groups are cached here)."""
        pass


class _SelfComm(TorchComm):
    """A size-1 sub-communicator inside a larger job (e.g. the column group of a p_c = 1 grid)."""

    def __init__(self, world_rank):
        self.group, self.world_rank, self.ranks, self.rank, self.size = None, world_rank, [world_rank], 0, 1
        self.device, self.backend = torch.device("cpu"), None

    def _solo(self):
        return True


class SoloGrid:
    """A 1 x 1 grid of this process alone inside a larger torch.distributed job: what the fits of a perturbation-shared NMFk
    sweep run on (pyDNMFk.PyNMFk, `params.nmfk_split = 'perturbations'`) -- every collective returns at once."""

    def __init__(self, world_rank=0):
        self.comm = _SelfComm(world_rank)
        self.rank, self.size, self.p_r, self.p_c, self.coord2d = 0, 1, 1, 1, [0, 0]

    def cart_1d_row(self):
        return self.comm

    def cart_1d_column(self):
        return self.comm

    def Free(self):
        pass
