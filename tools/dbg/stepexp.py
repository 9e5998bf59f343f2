"""Whole MU/FRO step with the fused W phase (dnmf_aht_update_w) vs two launches (dnmf_aht + dnmf_mu_update_w)."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pydnmfk_amd.engine import HIP_OPS as ops, new_gram
m = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
n, k = 8192, int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
A = torch.rand(m, n, device=dev, generator=g); W = torch.rand(m, k, device=dev, generator=g); H = torch.rand(k, n, device=dev, generator=g)
G = new_gram(k, dev); AtW = torch.empty(k, n, device=dev); AH = torch.empty(m, k, device=dev)
eps = 1.19e-7
def step(fusedw):
    ops.gram_hht(H, G)
    if fusedw: ops.aht_update_w(A, H, G, W, eps)
    else:
        ops.aht(A, H, AH); ops.mu_update_w(W, AH, G, eps)
    ops.gram_wtw(W, G); ops.wta(A, W, AtW); ops.mu_update_h(H, AtW, G, eps, False)
def timeit(fusedw, steps):
    for _ in range(10): step(fusedw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step(fusedw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
steps = 400 if m <= 65536 else 150
out = {"m": m, "k": k}
for rep in range(2):
    out["fused_%d" % rep] = round(timeit(True, steps), 4)
    out["unfused_%d" % rep] = round(timeit(False, steps), 4)
print(json.dumps(out))
