"""HALS step timing: python tools/halsbench.py m n k"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pydnmfk_amd.dist_comm import MPI_comm
from pydnmfk_amd.dist_nmf import nmf_algorithms_1D
from pydnmfk_amd.utils import parse
m, n, k = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (262144, 8192, 16)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
A = torch.rand(m, n, device=dev, generator=g); W = torch.rand(m, k, device=dev, generator=g); H = torch.rand(k, n, device=dev, generator=g)
comms = MPI_comm(None, 1, 1)
p = parse(); p.comm1, p.comm, p.p_r, p.p_c, p.k, p.m, p.n = comms.comm, comms, 1, 1, k, m, n
p.row_comm, p.col_comm = comms.cart_1d_row(), comms.cart_1d_column()
p.norm, p.W_update, p.eps = "fro", True, 1.1920929e-07
out = {}
for method in ("mu", "hals"):
    p.method = method
    for i in range(3): nmf_algorithms_1D(A, W, H, params=p).update()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(10): nmf_algorithms_1D(A, W, H, params=p).update()
    torch.cuda.synchronize(); out[method + "_ms"] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
print(json.dumps(out))
# the W sweep alone: persistent launch vs one launch per column
from pydnmfk_amd.engine import HIP_OPS as ops, new_gram
G = ops.gram_hht(H, new_gram(k, dev)); AH = torch.rand(m, k, device=dev, generator=g) * 100
def tm(fn, reps=10, warm=3):
    for _ in range(warm): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    xs = sorted(s.elapsed_time(e) for s, e in ev)
    return xs[len(xs) // 2]
Wc = W.clone()
out2 = {"w_sweep_persistent_ms": round(tm(lambda: ops.hals_update_w(Wc, AH, G, 1.19e-7)), 4),
        "w_sweep_columns_ms": round(tm(lambda: ops.hals_update_w_columns(Wc, AH, G, 1.19e-7)), 4),
        "stream_2mk4_ms_at_5TBs": round(2.0 * m * k * 4 / 5e9, 4)}
print(json.dumps(out2))
