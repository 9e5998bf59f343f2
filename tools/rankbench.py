"""What ONE rank of an N-GPU run does, measured on a one-GPU box: python tools/rankbench.py [--ranks 8] [--steps 300]

The rank-0 shard of BASELINE config 3 under a 1D row grid p_r = N (m/N x 8192 rows of X, k = 64) is stepped with the
real RCCL calls of the path (`init_process_group("nccl", world_size=1)` + `TorchComm.always_collective`, as
tests/test_gpu_rccl.py does): the kernels, the launch / Python overhead between them and the fixed cost of each
collective call are real; what is missing is the wire time of the 2 MiB allreduce over xGMI.  Prints ms/step next to the
sum of the step's kernel times (HIP events around the fused single-GPU step on the same shard), i.e. an upper bound of the
strong-scaling efficiency the driver's 8-GPU run can show:  eff <= t(1 GPU, full X) / (N * t(rank shard)).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--rows", type=int, default=262144)
    ap.add_argument("--cols", type=int, default=8192)
    ap.add_argument("--rank", dest="k", type=int, default=64)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--norm", default="fro")
    ap.add_argument("--method", default="mu", choices=["mu", "hals"],
                    help="hals: the W sweep exchanges one 8-byte column norm per column (k allreduces per step on a row grid)")
    ap.add_argument("--no-collectives", action="store_true", help="same shard, single-GPU path (no RCCL calls)")
    ap.add_argument("--exchange", default="torch", choices=["torch", "native"],
                    help="torch: dist.all_reduce between the launches; native: the library's own RCCL communicator, one C call per step")
    ap.add_argument("--chunks", type=int, default=None, help="overlap chunks of the H phase (default: the path's own default)")
    ap.add_argument("--direct", action="store_true", help="with --exchange native: the peer regions of the direct allreduce are set up (this one "
                    "rank is its own peer) -- the HALS W sweep then runs as ONE persistent launch whose column norms cross the ranks through "
                    "the slot slabs in those regions (no wire here: the kernel's own cost)")
    a = ap.parse_args()
    from pydnmfk_amd.dist_comm import MPI_comm, TorchComm
    from pydnmfk_amd.dist_nmf import nmf_algorithms_1D
    from pydnmfk_amd.utils import parse

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if not a.no_collectives:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=0, world_size=1)
        TorchComm.always_collective = True
    m_l, n, k = a.rows // a.ranks, a.cols, a.k
    comms = MPI_comm(None, 1, 1)
    p = parse()
    p.comm1, p.comm, p.k, p.m, p.n = comms.comm, comms, k, a.rows, n
    p.p_r, p.p_c = (1, 1) if a.no_collectives else (a.ranks, 1)
    p.row_comm, p.col_comm = comms.cart_1d_row(), comms.cart_1d_column()
    p.norm, p.method, p.W_update, p.eps = a.norm, a.method, True, 1.1920929e-07
    p.hals_force_exchange = True          # (one rank standing in for a row grid: the column norms still go through the collective)
    if a.chunks is not None:
        p.overlap_chunks = a.chunks
    if a.exchange == "native" and not a.no_collectives:
        from pydnmfk_amd.engine import NativeComm
        p.exchange, p.native_always = "native", True
        p._native_comm = NativeComm(comms.comm, 1, 1)
        p._native_comm.set_always_exchange(True)
        if a.direct:
            kp = 32 if k <= 32 else (64 if k <= 64 else 128)
            assert p._native_comm.enable_direct(comms.comm, k * max(m_l, n) + kp * kp + 4096)
            p._native_comm.set_direct(True)
    g = torch.Generator(device=dev).manual_seed(1)
    A = torch.rand(m_l, n, device=dev, generator=g)
    W = torch.rand(m_l, k, device=dev, generator=g)
    H = torch.rand(k, n, device=dev, generator=g)

    def step(i):
        nmf_algorithms_1D(A, W, H, params=p).update(clamp=(i % 10 == 0))

    for i in range(10):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    t_issue = time.perf_counter() - t0           # host time to ISSUE the steps (the GPU may lag behind)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    out = {"ranks_emulated": a.ranks, "rows_per_rank": m_l, "n": n, "k": k, "collectives": not a.no_collectives,
           "exchange": a.exchange, "chunks": a.chunks, "method": a.method, "direct": bool(a.direct),
           "ms_per_step": round(el / a.steps * 1e3, 4), "host_issue_ms_per_step": round(t_issue / a.steps * 1e3, 4)}
    print(json.dumps(out))
    if not a.no_collectives:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
