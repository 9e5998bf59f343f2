"""The direct (two-shot) allreduce over IPC peer buffers (csrc/dnmf_comm.hip, include/dnmf.h dnmf_comm_direct_*) with 2 and 4 ranks
stacked on the one GPU: every rank exports a region, gloo carries the handles, every rank maps its peers' regions.  Checked: the
result equals the rank-ordered float32 sum bit for bit on every rank (one owner per element), for message sizes from 2 floats to
the packed message of BASELINE config 3, repeated calls (the parity scheme), and a whole 1D MU step sequenced inside the library
(dnmf_mu_fro_step_1d over a hosted communicator whose world allreduce goes direct) equal to the same step over the hosted
transport alone.  What a one-GPU box cannot show is the xGMI wire: bench.py times the direct form against RCCL in its warm-up."""
import os
import traceback

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.multiprocessing as mp  # noqa: E402

pytestmark = pytest.mark.gpu


def _rank(rank, world, port, q):
    try:
        import torch.distributed as dist
        from pydnmfk_amd.dist_comm import MPI_comm
        from pydnmfk_amd.engine import NativeComm, _torch_hosted_collective
        torch.cuda.set_device(0)
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        comms = MPI_comm(None, world, 1)
        groups = {0: comms.comm, 1: comms.cart_1d_row(), 2: comms.cart_1d_column()}
        nc = NativeComm.hosted(world, rank, world, 1, _torch_hosted_collective(groups))
        n_max = 64 * 8192 + 64 * 64
        # every rank must size its region alike (peers compute offsets into each other's regions from ONE capacity): a mismatch
        # fails the set-up on EVERY rank, the regions are released, and a second set-up starts from scratch
        assert not nc.enable_direct(comms.comm, n_max + 64 * rank)
        assert "sized its region" in str(nc.direct_error)
        assert nc.enable_direct(comms.comm, n_max)
        assert nc.direct_self_check(comms.comm)            # first contact: direct sum == the communicator's own sum, agreed over ranks
        dev = torch.device("cuda", 0)
        out = {}
        for n in (2, 130, 4096, 100002, n_max):
            for rep in range(3):                                    # repeated calls: both parities, flags from earlier calls
                g = torch.Generator(device=dev)
                g.manual_seed(1000 * n + 10 * rep + rank)
                x = torch.rand(n, device=dev, generator=g) - 0.3
                parts = comms.comm.allgather_blocks(x, [(n,)] * world)
                ref = parts[0].clone()
                for p in parts[1:]:
                    ref += p                                         # rank-ordered float32 sum
                y = nc.allreduce_direct_(x.clone())
                torch.cuda.synchronize()
                out[(n, rep)] = bool(torch.equal(y, ref))
        # the one-launch form for a few doubles (the HALS column norms): rank-ordered float64 sum, bit for bit
        for cnt in (1, 3, 8):
            for rep in range(4):
                g = torch.Generator(device=dev)
                g.manual_seed(77 * cnt + rep + 1000 * rank)
                x = torch.rand(cnt, device=dev, generator=g, dtype=torch.float64) * 1e3
                parts = comms.comm.allgather_blocks(x.float(), [(cnt,)] * world)      # (transport for the check only; fp32 view)
                xs = [None] * world
                import torch.distributed as dist2
                dist2.all_gather_object(xs, x.cpu())
                ref = xs[0].clone()
                for p in xs[1:]:
                    ref += p
                y = nc.allreduce_direct_f64_(x.clone())
                torch.cuda.synchronize()
                out[("f64", cnt, rep)] = bool(torch.equal(y.cpu(), ref))
        assert not nc.direct_timed_out()
        # a whole 1D step inside the library: the packed exchange through the direct path vs through the hosted transport
        rs = np.random.RandomState(5)
        m, n, k = 512, 384, 16
        A = np.abs(rs.rand(m, n)).astype(np.float32)
        W0, H0 = rs.rand(m, k).astype(np.float32), rs.rand(k, n).astype(np.float32)
        r0, r1 = rank * (m // world), (rank + 1) * (m // world)
        res = []
        for direct in (False, True):
            nc.set_direct(direct)
            dA, dW, dH = (torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in (A[r0:r1], W0[r0:r1], H0))
            for it in range(3):
                nc.step_1d("fro", dA, dW, dH, 1.1920929e-07, True, it == 0)
            torch.cuda.synchronize()
            res.append((dW.cpu(), dH.cpu()))
        if world == 2:       # a + b in either order: the same bits as over the hosted transport
            out["step"] = bool(torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]))
        else:                # gloo sums four ranks in its own order, the direct form in rank order: equal to fp32 rounding
            out["step"] = bool(torch.allclose(res[0][0], res[1][0], rtol=2e-6, atol=1e-7) and torch.allclose(res[0][1], res[1][1], rtol=2e-6, atol=1e-7))
        # HALS: the k column norms of the W sweep go through the one-launch form when direct is on
        res = []
        for direct in (False, True):
            nc.set_direct(direct)
            before = nc.hals_xsweeps()
            dA, dW, dH = (torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in (A[r0:r1], W0[r0:r1], H0))
            for it in range(3):
                nc.hals_step_1d(dA, dW, dH, 1.1920929e-07, True, it == 0)
            torch.cuda.synchronize()
            res.append((dW.cpu(), dH.cpu()))
            out["xsweeps_%s" % direct] = (nc.hals_xsweeps() - before) == (3 if direct else 0)    # one persistent launch per W sweep
            from pydnmfk_amd.engine import HIP_OPS
            HIP_OPS.hals_check()                             # (no wait of the sweep gave up)
            if direct:                                       # every rank ends with IDENTICAL H (the slots are summed in one order everywhere)
                hs = [None] * world
                dist.all_gather_object(hs, dH.cpu())
                out["identical_H"] = all(torch.equal(hs[0], h) for h in hs[1:])
        # (round 5: with the peer regions up the W sweep is ONE persistent launch whose column norms cross the ranks through the slot
        #  slabs in those regions -- a different evaluation of the same sweep than the column launches of the hosted transport
        #  (csrc/dnmf_hals.h), and a HALS sweep amplifies the last bits of every norm: its per-step parity budget is 5e-5, tests/_mp.py)
        tol = dict(rtol=2e-4, atol=2e-5)
        out["hals_step"] = bool(torch.allclose(res[0][0], res[1][0], **tol) and torch.allclose(res[0][1], res[1][1], **tol))
        # a peer that never arrives: the wait gives up after the configured time and says so (sticky status word) instead of
        # hanging the GPU.  Last use of the regions in this test: the sequence numbers of the ranks differ afterwards.
        dist.barrier()
        nc.set_direct_timeout(0.2)
        if rank == 0:
            nc.allreduce_direct_(torch.ones(64, device=dev))
            torch.cuda.synchronize()
            out["timeout_reported"] = nc.direct_timed_out()
        else:
            out["timeout_reported"] = not nc.direct_timed_out()
        q.put((rank, out, None))
        dist.barrier()
        nc.close()
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        q.put((rank, None, traceback.format_exc()))


@pytest.mark.parametrize("world", [2, 4])
def test_direct_allreduce_on_stacked_ranks(world):
    from tests._mp import free_port
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_rank, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, out, err in res:
        assert err is None, "rank %d failed:\n%s" % (rank, err)
        bad = [key for key, ok in out.items() if not ok]
        assert not bad, (rank, bad)
