"""Compute-side cost of the chunked / overlapped H phase (dist_nmf._fro_h_phase_overlapped) on ONE GPU: the per-rank step of
the p_r = 8 configuration (32768 x 8192, k = 64) with the H phase in 1 / 2 / 4 / 8 column chunks and no exchange at all
(a one-rank communicator).  What the chunks cost here is what the overlapped allreduce has to win back on 8 GPUs."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pydnmfk_amd.dist_comm import MPI_comm
from pydnmfk_amd.dist_nmf import nmf_algorithms_1D
from pydnmfk_amd.utils import parse
m, n, k = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (32768, 8192, 64)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
A = torch.rand(m, n, device=dev, generator=g); W = torch.rand(m, k, device=dev, generator=g); H = torch.rand(k, n, device=dev, generator=g)
comms = MPI_comm(None, 1, 1)
p = parse(); p.comm1, p.comm, p.p_r, p.p_c, p.k, p.m, p.n = comms.comm, comms, 8, 1, k, m * 8, n   # pretends to be row 0 of 8 x 1
p.row_comm, p.col_comm = comms.cart_1d_row(), comms.cart_1d_column()
p.norm, p.method, p.W_update, p.eps = "fro", "mu", True, 1.1920929e-07
out = {}
for nch in (1, 2, 4, 8):
    p.overlap_chunks = nch
    for i in range(5): nmf_algorithms_1D(A, W, H, params=p).update(clamp=(i % 10 == 0))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(100): nmf_algorithms_1D(A, W, H, params=p).update(clamp=(i % 10 == 0))
    torch.cuda.synchronize(); out["chunks_%d_ms" % nch] = round((time.perf_counter() - t0) / 100 * 1e3, 4)
print(json.dumps(out))
