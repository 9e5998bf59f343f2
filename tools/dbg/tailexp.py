"""How long is the fused W-phase kernel when the contraction is short (the main loop nearly empty)?  -> its fixed tail."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pydnmfk_amd.engine import HIP_OPS as ops, new_gram
dev = torch.device("cuda", 0)
def t(fn, reps=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    x = sorted(s.elapsed_time(e) for s, e in ev)
    return x[len(x)//2]
for m in (32768, 262144):
    for n in (64, 256, 1024):
        for k in (64,):
            A = torch.rand(m, n, device=dev); W = torch.rand(m, k, device=dev); H = torch.rand(k, n, device=dev)
            G = ops.gram_hht(H, new_gram(k, dev)); V = torch.empty(m, k, device=dev)
            print(json.dumps({"m": m, "n": n, "k": k, "fused_us": round(1e3 * t(lambda: ops.aht_update_w(A, H, G, W, 1.19e-7)), 1),
                              "aht_us": round(1e3 * t(lambda: ops.aht(A, H, V)), 1), "update_w_us": round(1e3 * t(lambda: ops.mu_update_w(W, V, G, 1.19e-7)), 1)}))
