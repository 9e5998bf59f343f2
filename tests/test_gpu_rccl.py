"""The RCCL transport on a one-GPU box: `init_process_group("nccl", world_size=1)` in a spawned child, with
`TorchComm.always_collective` switched on so that single-rank communicators still issue the real torch.distributed
calls.  What this proves: the nccl code path of pydnmfk_amd.dist_comm loads, creates sub-groups, takes device buffers of
every dtype the path exchanges (float32 packed products, float64 norms, int64 shapes), stages CPU tensors through the
device, and is stream-ordered against the HIP kernels (kernel -> collective -> kernel sequences give the single-rank
result).  What it cannot show is a second rank: scaling over xGMI stays unmeasured until the driver has an 8-GPU node."""
import traceback

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _child(port, q):
    try:
        import os
        import torch.distributed as dist
        from oracle import nmf_oracle as orc
        from pydnmfk_amd.dist_comm import MPI_comm, TorchComm
        from pydnmfk_amd.dist_nmf import nmf_algorithms_1D, nmf_algorithms_2D
        from pydnmfk_amd.engine import HIP_OPS as ops, new_gram
        from pydnmfk_amd.pyDNMF import PyNMF
        from pydnmfk_amd.utils import parse
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", rank=0, world_size=1)
        TorchComm.always_collective = True
        world = TorchComm(None)
        assert world.backend == "nccl" and world.device.type == "cuda" and not world._solo()
        log = {}
        # --- every collective of the path on device buffers (1 rank: SUM / gather = identity)
        x = torch.rand(64 * 8192 + 64 * 64, device=dev)                      # the packed [W^T A | W^T W] buffer of config 3
        ref = x.clone()
        world.allreduce_(x)
        assert torch.equal(x, ref)
        d = torch.tensor([1.5, 2.5], dtype=torch.float64, device=dev)
        world.allreduce_(d)
        assert d.tolist() == [1.5, 2.5]
        assert world.allreduce(7) == 7 and world.allreduce(2.5) == 2.5       # int64 / float64 scalars (compute_global_dim)
        assert np.array_equal(world.allreduce(np.arange(5.0)), np.arange(5.0))
        c = torch.rand(3, 4)                                                 # CPU tensor: staged through the device
        assert torch.equal(world.allreduce(c), c) and not world.allreduce(c).is_cuda
        blk = torch.rand(5, 7, device=dev)
        g = world.allgather_blocks(blk, [(5, 7)])
        assert len(g) == 1 and torch.equal(g[0], blk)
        rs = world.reduce_scatter_rows(blk, [5])
        assert torch.equal(rs, blk)
        assert torch.equal(world.bcast(blk, root=0), blk)
        assert world.allgather(3) == [3]
        world.barrier()
        # --- sub-groups (dist.new_group under nccl) and a collective on them
        sub = TorchComm(dist.new_group([0]), [0])
        y = torch.rand(1000, device=dev)
        yr = y.clone()
        sub.allreduce_(y)
        assert torch.equal(y, yr)
        comms = MPI_comm(None, 1, 1)
        assert comms.cart_1d_row().size == 1 and comms.cart_1d_column().size == 1
        # --- HIP kernels and collectives interleaved on the stream: the 1D-row exchange sequence of dist_nmf.py
        rsn = np.random.RandomState(3)
        m, n, k = 4096, 1024, 64
        A = np.abs(rsn.rand(m, k) @ rsn.rand(k, n) + 0.01 * rsn.randn(m, n)).astype(np.float32)
        W0, H0 = rsn.rand(m, k).astype(np.float32), rsn.rand(k, n).astype(np.float32)
        eps = float(np.finfo(np.float32).eps)
        Wr, Hr = orc.fro_mu_step_local(A, W0.copy(), H0.copy(), np.float32(eps))
        Ad, Wd, Hd = (torch.from_numpy(v).to(dev) for v in (A, W0, H0))
        G = ops.gram_hht(Hd, new_gram(k, dev))
        ops.aht_update_w(Ad, Hd, G, Wd, eps)
        buf = torch.zeros(k * n + 64 * 64, device=dev)
        AtW, G2 = buf[: k * n].view(k, n), buf[k * n:].view(64, 64)
        for _ in range(3):                                                   # kernel -> RCCL -> kernel, repeatedly
            ops.gram_wtw(Wd, G2)
            ops.wta(Ad, Wd, AtW)
            world.allreduce_(buf)
        ops.mu_update_h(Hd, AtW, G2, eps, False)
        log["dW"] = float(np.linalg.norm(Wd.cpu().numpy() - Wr) / np.linalg.norm(Wr))
        log["dH"] = float(np.linalg.norm(Hd.cpu().numpy() - Hr) / np.linalg.norm(Hr))
        # --- the 2D choreography (allgather / reduce-scatter on sub-groups) with one-rank groups == the 1D step
        args = parse()
        args.comm1, args.comm, args.p_r, args.p_c, args.k, args.m, args.n = world, comms, 1, 1, k, m, n
        args.row_comm, args.col_comm = TorchComm(dist.new_group([0]), [0]), TorchComm(dist.new_group([0]), [0])
        args.eps, args.W_update, args.norm, args.method = eps, True, "fro", "mu"
        for norm in ("fro", "kl"):
            args.norm = norm
            W1, H1 = torch.from_numpy(W0).to(dev), torch.from_numpy(H0).to(dev)
            W2, H2 = W1.clone(), H1.clone()
            TorchComm.always_collective = False
            nmf_algorithms_1D(Ad, W1, H1, params=args).update()
            TorchComm.always_collective = True
            nmf_algorithms_2D(Ad, W2, H2, params=args).update()
            log["2d_%s" % norm] = (float((W1 - W2).norm() / W1.norm()), float((H1 - H2).norm() / H1.norm()))
        # --- the overlapped H phase of row grids with more than two ranks: column chunks whose allreduce is started with
        #     async_op=True on RCCL's stream while the next chunk is computed.  One rank pretending to be row 0 of a 4 x 1
        #     grid: the sums over a one-rank group are identities, so the step must equal the plain single-rank step.
        args4 = parse()
        args4.comm1, args4.comm, args4.p_r, args4.p_c, args4.k, args4.m, args4.n = world, comms, 4, 1, k, m, n
        args4.eps, args4.W_update, args4.norm, args4.method = eps, True, "fro", "mu"
        args4.overlap_min_cols, args4.overlap_chunks = 128, 4
        W5, H5 = torch.from_numpy(W0).to(dev), torch.from_numpy(H0).to(dev)
        alg = nmf_algorithms_1D(Ad, W5, H5, params=args4)
        assert alg._overlap_chunks(n) == 4
        for i in range(3):
            nmf_algorithms_1D(Ad, W5, H5, params=args4).update(clamp=(i == 0))
        W6, H6 = torch.from_numpy(W0).to(dev), torch.from_numpy(H0).to(dev)
        args1 = parse()
        args1.comm1, args1.comm, args1.p_r, args1.p_c, args1.k, args1.m, args1.n = world, comms, 1, 1, k, m, n
        args1.eps, args1.W_update, args1.norm, args1.method = eps, True, "fro", "mu"
        TorchComm.always_collective = False
        for i in range(3):
            nmf_algorithms_1D(Ad, W6, H6, params=args1).update(clamp=(i == 0))
        TorchComm.always_collective = True
        log["overlap"] = (float((W5 - W6).norm() / W6.norm()), float((H5 - H6).norm() / H6.norm()))
        # --- the exchange INSIDE the library (csrc/dnmf_comm.hip, include/dnmf.h "Grid exchanges"): libdnmf_hip.so creates its
        #     own RCCL communicator from a broadcast id and runs whole 1D steps -- kernels -> ncclAllReduce -> kernels, the
        #     chunked exchange on its internal stream -- in ONE call.  Must equal the Python choreography bit for bit
        #     (same kernels, same order, same buffers), for the packed allreduce and for 2 / 4 overlapped chunks, FRO and KL.
        from pydnmfk_amd.engine import NativeComm
        nc = NativeComm(world, 1, 1)
        nc.set_always_exchange(True)
        xx = torch.rand(70000, device=dev)
        xx0 = xx.clone()
        nc.allreduce_(xx)
        assert torch.equal(xx, xx0)
        native = {}
        for norm in ("fro", "kl"):
            for chunks in ((1, 2, 4) if norm == "fro" else (1,)):
                ap = parse()
                ap.comm1, ap.comm, ap.p_r, ap.p_c, ap.k, ap.m, ap.n = world, comms, 4, 1, k, m, n
                ap.eps, ap.W_update, ap.norm, ap.method = eps, True, norm, "mu"
                ap.overlap_min_cols, ap.overlap_chunks = 128, chunks
                an = parse()
                an.__dict__.update(vars(ap))
                an.exchange, an.native_always, an._native_comm = "native", True, nc
                Wp, Hp = torch.from_numpy(W0).to(dev), torch.from_numpy(H0).to(dev)
                Wn, Hn = Wp.clone(), Hp.clone()
                for i in range(3):
                    nmf_algorithms_1D(Ad, Wp, Hp, params=ap).update(clamp=(i == 0))
                    nmf_algorithms_1D(Ad, Wn, Hn, params=an).update(clamp=(i == 0))
                assert nc.overlap_chunks == chunks
                native["%s_%d" % (norm, chunks)] = bool(torch.equal(Wp, Wn) and torch.equal(Hp, Hn))
        # the same on shapes the tiles do not divide (ragged rows / columns, ranks below the padded width, a rank on the
        # 16-wide kernels, W_update = False): the C step's workspace layout and chunk arithmetic against the Python one
        odd = True
        for (mm, nn, kk, nrm, chunks, wupd) in [(1000, 260, 5, "fro", 2, True), (257, 131, 33, "kl", 1, True), (3000, 777, 16, "fro", 4, True),
                                                (130, 8200, 128, "fro", 2, False), (513, 640, 16, "kl", 1, True), (64, 4100, 7, "fro", 4, True)]:
            rq = np.random.RandomState(mm + nn + kk)
            Aq = torch.from_numpy(rq.rand(mm, nn).astype(np.float32)).to(dev)
            Wq0, Hq0 = rq.rand(mm, kk).astype(np.float32), rq.rand(kk, nn).astype(np.float32)
            ap = parse()
            ap.comm1, ap.comm, ap.p_r, ap.p_c, ap.k, ap.m, ap.n = world, comms, 4, 1, kk, mm, nn
            ap.eps, ap.W_update, ap.norm, ap.method = eps, wupd, nrm, "mu"
            ap.overlap_min_cols, ap.overlap_chunks = 64, chunks
            an = parse()
            an.__dict__.update(vars(ap))
            an.exchange, an.native_always, an._native_comm = "native", True, nc
            Wp, Hp = torch.from_numpy(Wq0).to(dev), torch.from_numpy(Hq0).to(dev)
            Wn, Hn = Wp.clone(), Hp.clone()
            for i in range(2):
                nmf_algorithms_1D(Aq, Wp, Hp, params=ap).update(clamp=(i == 0))
                nmf_algorithms_1D(Aq, Wn, Hn, params=an).update(clamp=(i == 0))
            odd = odd and bool(torch.equal(Wp, Wn) and torch.equal(Hp, Hn))
        native["odd_shapes"] = odd
        # --- the 2D entry points (dnmf_mu_{fro,kl}_step_2d: allreduce + allgather + reduce-scatter inside the library) on the
        #     one-rank communicator == the Python 2D choreography bit for bit; shapes on the 16-wide and 32-wide kernels, W_update
        #     False, clamp, column counts that are not whole vectors; slices off the grid's partition rule are refused
        two_d = True
        for (mm, nn, kk, nrm, wupd) in [(4096, 1024, 64, "fro", True), (4096, 1024, 64, "kl", True), (1000, 260, 5, "fro", True),
                                        (513, 640, 16, "kl", True), (300, 128, 33, "kl", False), (2049, 512, 128, "fro", False),
                                        (64, 130, 8, "fro", True), (257, 131, 17, "kl", True)]:      # columns that are not whole vectors
            rq = np.random.RandomState(mm + nn + kk + 1)
            Aq = torch.from_numpy(rq.rand(mm, nn).astype(np.float32)).to(dev)
            Wq0, Hq0 = rq.rand(mm, kk).astype(np.float32), rq.rand(kk, nn).astype(np.float32)
            ap = parse()
            ap.comm1, ap.comm, ap.p_r, ap.p_c, ap.k, ap.m, ap.n = world, comms, 1, 1, kk, mm, nn
            ap.row_comm, ap.col_comm = args.row_comm, args.col_comm
            ap.eps, ap.W_update, ap.norm, ap.method = eps, wupd, nrm, "mu"
            an = parse()
            an.__dict__.update(vars(ap))
            an.exchange, an._native_comm = "native", nc
            Wp, Hp = torch.from_numpy(Wq0).to(dev), torch.from_numpy(Hq0).to(dev)
            Wn, Hn = Wp.clone(), Hp.clone()
            for i in range(2):
                nmf_algorithms_2D(Aq, Wp, Hp, params=ap).update(clamp=(i == 0))
                alg = nmf_algorithms_2D(Aq, Wn, Hn, params=an)
                assert alg._native_step.__func__ is nmf_algorithms_2D._native_step and nc.step_2d_ok(Aq, Wn, Hn)
                alg.update(clamp=(i == 0))
            two_d = two_d and bool(torch.equal(Wp, Wn) and torch.equal(Hp, Hn))
        native["2d"] = two_d
        # --- HALS / Frobenius in the library (dnmf_hals_fro_step_1d): with `always` the one-rank communicator runs the exchanged
        #     W sweep -- column kernel, ncclAllReduce of ONE double, next column -- which must equal the column form of the
        #     choreography bit for bit (the same column kernels; a one-rank sum is the identity)
        hals = True
        for (mm, nn, kk) in [(1000, 260, 5), (4096, 1024, 64), (513, 640, 16)]:
            rq = np.random.RandomState(mm + nn + kk + 2)
            Aq = torch.from_numpy(np.abs(rq.rand(mm, kk) @ rq.rand(kk, nn) + 0.01 * rq.randn(mm, nn)).astype(np.float32)).to(dev)
            Wq0, Hq0 = rq.rand(mm, kk).astype(np.float32), rq.rand(kk, nn).astype(np.float32)
            ap = parse()
            ap.comm1, ap.comm, ap.p_r, ap.p_c, ap.k, ap.m, ap.n = world, comms, 4, 1, kk, mm, nn
            ap.eps, ap.W_update, ap.norm, ap.method, ap.hals_sweep = eps, True, "fro", "hals", "columns"
            an = parse()
            an.__dict__.update(vars(ap))
            an.exchange, an.native_always, an._native_comm = "native", True, nc
            Wp, Hp = torch.from_numpy(Wq0).to(dev), torch.from_numpy(Hq0).to(dev)
            Wn, Hn = Wp.clone(), Hp.clone()
            TorchComm.always_collective = False
            for i in range(2):
                nmf_algorithms_1D(Aq, Wp, Hp, params=ap).update(clamp=(i == 0))
            TorchComm.always_collective = True
            for i in range(2):
                nmf_algorithms_1D(Aq, Wn, Hn, params=an).update(clamp=(i == 0))
            hals = hals and bool(torch.equal(Wp, Wn) and torch.equal(Hp, Hn))
        native["hals_1d"] = hals
        Aq = torch.rand(64, 130, device=dev)                      # a W slice that is not the grid's share of the block's rows
        Wq, Hq = torch.rand(63, 8, device=dev), torch.rand(8, 130, device=dev)
        assert not nc.step_2d_ok(Aq, Wq, Hq)
        try:
            nc.step_2d("fro", Aq, Wq, Hq, eps)
            native["2d_refuses_uneven"] = False
        except Exception as ex:  # noqa: BLE001
            native["2d_refuses_uneven"] = "host choreography" in str(ex)
        log["native"] = native
        nc.close()
        # --- a whole fit with the nccl group up (relative_err allreduces a float64 pair on the device)
        args2 = parse()
        args2.comm1, args2.comm, args2.p_r, args2.p_c, args2.k = world, comms, 1, 1, k
        args2.row_comm, args2.col_comm = comms.cart_1d_row(), comms.cart_1d_column()
        args2.itr, args2.init, args2.verbose, args2.prune, args2.norm, args2.method = 12, "rand", False, False, "fro", "mu"
        Wf, Hf, err = PyNMF(A, factors=[W0, H0], params=args2).fit()
        Wo, Ho, erro = orc.fit_single(A, W0, H0, 12, norm="fro")
        log["fit"] = (float(np.linalg.norm(Wf - Wo) / np.linalg.norm(Wo)), float(np.linalg.norm(Hf - Ho) / np.linalg.norm(Ho)),
                      abs(err - erro))
        dist.barrier(device_ids=[0])
        dist.destroy_process_group()
        q.put((log, None))
    except Exception:  # noqa: BLE001
        q.put((None, traceback.format_exc()))


def test_rccl_code_path_on_one_gpu():
    import torch.multiprocessing as mp
    from tests._mp import free_port
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_child, args=(free_port(), q))
    p.start()
    log, err = q.get(timeout=300)
    p.join(timeout=60)
    assert err is None, err
    assert log["dW"] <= 1e-5 and log["dH"] <= 1e-5, log
    for norm in ("fro", "kl"):
        assert max(log["2d_%s" % norm]) <= 1e-6, log
    assert max(log["overlap"]) <= 2e-6, log
    assert log["fit"][0] <= 1e-4 and log["fit"][1] <= 1e-4 and log["fit"][2] <= 1e-5, log
    assert log["native"] == {"fro_1": True, "fro_2": True, "fro_4": True, "kl_1": True, "odd_shapes": True, "2d": True,
                             "hals_1d": True, "2d_refuses_uneven": True}, log
