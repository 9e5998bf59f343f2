"""U H^T alone: python tools/uhtbench.py m n k   (tuning build: DNMF_KLUHT_VAR / DNMF_KLUHT_ABL / DNMF_KLUHT_PIPE select variants)"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pydnmfk_amd.engine import HIP_OPS as ops
m, n, k = (int(x) for x in sys.argv[1:4])
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
A = torch.rand(m, n, device=dev, generator=g); W = torch.rand(m, k, device=dev, generator=g); H = torch.rand(k, n, device=dev, generator=g)
UHT = torch.empty(m, k, device=dev); WTU = torch.empty(k, n, device=dev)
which = os.environ.get("KB", "uht")
fn = (lambda: ops.kl_uht(A, W, H, 1.19e-7, UHT)) if which == "uht" else (lambda: ops.kl_wtu(A, W, H, 1.19e-7, WTU))
for _ in range(5): fn()
reps = int(os.environ.get("REPS", "30"))
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
for s, e in ev:
    s.record(); fn(); e.record()
torch.cuda.synchronize()
x = sorted(s.elapsed_time(e) for s, e in ev)
ms = x[len(x) // 2]
print(json.dumps({"which": which, "m": m, "n": n, "k": k, "var": os.environ.get("DNMF_KLUHT_VAR", ""), "abl": os.environ.get("DNMF_KLUHT_ABL", ""),
                  "ms": round(ms, 4), "min_ms": round(x[0], 4), "tf": round(4.0 * m * n * k / ms / 1e9, 1)}))
