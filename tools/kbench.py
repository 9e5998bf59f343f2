"""Quick per-kernel timing (HIP events) for A/B experiments: python tools_kbench.py [m n k]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pydnmfk_amd.engine import HIP_OPS as ops, new_gram
m, n, k = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (262144, 8192, 64)
if "--bf16" in sys.argv: os.environ["BF16"] = "1"
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
A = torch.rand(m, n, device=dev, generator=g) if not os.environ.get("ALIAS") else torch.rand(1, n, device=dev, generator=g).expand(m, n); W = torch.rand(m, k, device=dev, generator=g); H = torch.rand(k, n, device=dev, generator=g)
if os.environ.get("BF16"): A = A.to(torch.bfloat16)   # bf16 storage of A
abytes = A.element_size() * m * n
G = ops.gram_hht(H, new_gram(k, dev)); AtW = torch.empty(k, n, device=dev); Wt = W.clone()
def t(fn, reps=8, warm=3):
    for _ in range(warm): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    x = sorted(s.elapsed_time(e) for s, e in ev)
    return x[len(x)//2]
which = os.environ.get("KB", "nt,tn")
out = {}
if "nt" in which:
    ms = t(lambda: ops.aht_update_w(A, H, G, Wt, 1.19e-7)); out["nt_ms"] = round(ms, 4); out["nt_tf"] = round((2.0*m*n*k + 2.0*m*k*k)/ms/1e9, 1); out["nt_tbs"] = round(abytes/ms/1e9, 2)
if "tn" in which:
    ms = t(lambda: ops.wta(A, W, AtW)); out["tn_ms"] = round(ms, 4); out["tn_tf"] = round(2.0*m*n*k/ms/1e9, 1); out["tn_tbs"] = round(abytes/ms/1e9, 2)
if "norm" in which:
    ms = t(lambda: ops.sqnorm(A)); out["sq_ms"] = round(ms, 4); out["sq_tbs"] = round(abytes/ms/1e9, 2)
    ms = t(lambda: ops.resid_sqnorm(A, W, H)); out["res_ms"] = round(ms, 4); out["res_tbs"] = round(abytes/ms/1e9, 2)
print(json.dumps(out))
