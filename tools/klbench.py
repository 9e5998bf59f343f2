"""KL kernel timing: python tools/klbench.py m n k"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pydnmfk_amd.engine import HIP_OPS as ops
m, n, k = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (32768, 32768, 128)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
A = torch.rand(m, n, device=dev, generator=g); W = torch.rand(m, k, device=dev, generator=g); H = torch.rand(k, n, device=dev, generator=g)
UHT = torch.empty(m, k, device=dev); WTU = torch.empty(k, n, device=dev)
def t(fn, reps=int(os.environ.get("REPS", "10")), warm=3):
    for _ in range(warm): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    x = sorted(s.elapsed_time(e) for s, e in ev)
    return x[len(x)//2]
out = {}
fl = 4.0 * m * n * k
ms = t(lambda: ops.kl_uht(A, W, H, 1.19e-7, UHT)); out["uht_ms"] = round(ms, 3); out["uht_tf"] = round(fl/ms/1e9, 1)
ms = t(lambda: ops.kl_wtu(A, W, H, 1.19e-7, WTU)); out["wtu_ms"] = round(ms, 3); out["wtu_tf"] = round(fl/ms/1e9, 1)
ms = t(lambda: ops.resid_sqnorm(A, W, H)); out["resid_ms"] = round(ms, 3); out["resid_tf"] = round(2.0*m*n*k/ms/1e9, 1)
Wc, Hc = W.clone(), H.clone()
ms = t(lambda: ops.mu_kl_step(A, Wc, Hc, 1.19e-7, True, False), reps=3); out["kl_step_ms"] = round(ms, 3); out["kl_step_tf"] = round(2*fl/ms/1e9, 1)
print(json.dumps(out))
