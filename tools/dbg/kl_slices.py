"""The H-phase product of a 2D KL step on the config-4 block (32768^2, k = 128, p_r = 4): ONE full-width kl_wtu + a repack of the
k x n_l result into the reduce-scatter's [p_r][k][n_h] member blocks, against p_r sliced launches that write the blocks directly."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pydnmfk_amd.engine import HIP_OPS as ops
m, n, k, p = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (32768, 32768, 128, 4)
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1)
A = torch.rand(m, n, device=dev, generator=g); W = torch.rand(m, k, device=dev, generator=g); H = torch.rand(k, n, device=dev, generator=g)
nh = n // p
Hs = torch.stack([H[:, q * nh:(q + 1) * nh].contiguous() for q in range(p)])      # the allgather's receive buffer [p][k][nh]
Yb = torch.empty(p, k, nh, device=dev); Y = torch.empty(k, n, device=dev); Hj = torch.empty(k, n, device=dev)
eps = 1.1920929e-07
def t(fn, reps=10, warm=3):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
def sliced():
    for q in range(p):
        ops.kl_wtu(A[:, q * nh:(q + 1) * nh], W, Hs[q], eps, Yb[q])
def full():
    Hj.view(k, p, nh).copy_(Hs.permute(1, 0, 2))                 # assemble H_j from the member blocks
    ops.kl_wtu(A, W, Hj, eps, Y)
    Yb.copy_(Y.view(k, p, nh).permute(1, 0, 2))                  # repack into member blocks
def full_only():
    ops.kl_wtu(A, W, Hj, eps, Y)
out = {"shape": [m, n, k, p], "sliced_ms": t(sliced), "full_with_copies_ms": t(full), "full_only_ms": t(full_only)}
full(); a = Yb.clone(); sliced()
out["max_rel_diff"] = float((a - Yb).abs().max() / Yb.abs().max())
print(json.dumps(out))
