"""Fixed cost of the direct two-shot allreduce (csrc/dnmf_comm.hip, dnmf_comm_direct_*) with the ranks STACKED on one GPU:
copy + 4 launches + the flag handshakes, no xGMI wire -- the floor of what a real node adds its transfer time to.
  python tools/directbench.py [ranks=2] [floats=528384]      (528384 = the packed [W^T A | W^T W] message of BASELINE config 3)"""
import os, sys, json, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.multiprocessing as mp


def rank_main(rank, world, port, n, q):
    try:
        import torch.distributed as dist
        from pydnmfk_amd.dist_comm import MPI_comm
        from pydnmfk_amd.engine import NativeComm, _torch_hosted_collective
        torch.cuda.set_device(0)
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        comms = MPI_comm(None, world, 1)
        nc = NativeComm.hosted(world, rank, world, 1, _torch_hosted_collective({0: comms.comm, 1: comms.cart_1d_row(), 2: comms.cart_1d_column()}))
        assert nc.enable_direct(comms.comm, n)
        x = torch.rand(n, device="cuda")
        for _ in range(20):
            nc.allreduce_direct_(x)
        torch.cuda.synchronize(); dist.barrier()
        reps = 200
        t0 = time.perf_counter()
        for _ in range(reps):
            nc.allreduce_direct_(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps * 1e6
        y = torch.rand(1, device="cuda", dtype=torch.float64)                     # the HALS column norm: 8 bytes, one launch
        for _ in range(20):
            nc.allreduce_direct_f64_(y)
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            nc.allreduce_direct_f64_(y)
        torch.cuda.synchronize()
        dt8 = (time.perf_counter() - t0) / reps * 1e6
        q.put((rank, (dt, dt8), None))
        dist.barrier(); nc.close(); dist.destroy_process_group()
    except Exception:
        q.put((rank, None, traceback.format_exc()))


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 64 * 8192 + 64 * 64
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=rank_main, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs: p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs: p.join(timeout=30)
    for r, dt, err in res:
        assert err is None, err
    print(json.dumps({"ranks_stacked_on_one_gpu": world, "floats": n, "bytes": 4 * n, "us_per_allreduce_max_over_ranks": round(max(dt[0] for _, dt, _ in res), 1),
                      "us_per_8_byte_f64_allreduce": round(max(dt[1] for _, dt, _ in res), 1),
                      "note": "copy + 4 launches + 2 flag handshakes; every rank's kernels share the one GPU, no wire"}))
