import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pydnmfk_amd.engine import HIP_OPS as ops
m, n, k, p = 32768, 32768, 128, 4
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
A = torch.rand(m, n, device=dev, generator=g); W = torch.rand(m, k, device=dev, generator=g); H = torch.rand(k, n, device=dev, generator=g)
WTU = torch.empty(k, n, device=dev); blk = torch.empty(p, k, n // p, device=dev)
def t(fn, reps=5, warm=2):
    for _ in range(warm): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    x = sorted(s.elapsed_time(e) for s, e in ev); return x[len(x)//2]
def sliced():
    nh = n // p
    for q in range(p):
        ops.kl_wtu(A[:, q*nh:(q+1)*nh], W, H[:, q*nh:(q+1)*nh], 1.19e-7, blk[q])
def sliced_wta():
    nh = n // p
    for q in range(p):
        ops.wta(A[:, q*nh:(q+1)*nh], W, blk[q])
print(json.dumps({"kl_wtu_full_ms": round(t(lambda: ops.kl_wtu(A, W, H, 1.19e-7, WTU)), 3), "kl_wtu_4slices_ms": round(t(sliced), 3),
                  "wta_full_ms": round(t(lambda: ops.wta(A, W, WTU)), 3), "wta_4slices_ms": round(t(sliced_wta), 3)}))
