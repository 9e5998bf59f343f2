"""HBM-bound eltwise kernels in isolation on a long H (k=64, n=2^22 -> 3.2 GB per update)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pydnmfk_amd.engine import HIP_OPS as ops, new_gram
k = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 22
dev = torch.device("cuda", 0)
pad = int(os.environ.get("PAD", "0"))   # leading-dimension padding (floats): breaks the power-of-two row pitch
H = torch.rand(k, n + pad, device=dev)[:, :n]; S = torch.rand(k, n + pad, device=dev)[:, :n]; x = torch.rand(k, device=dev) + 1
W = torch.rand(n, k, device=dev); SW = torch.rand(n, k, device=dev)
G = new_gram(k, dev); G[:k, :k] = torch.rand(k, k, device=dev)
def t(fn, reps=20, warm=10):
    for _ in range(warm): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    xs = sorted(s.elapsed_time(e) for s, e in ev)
    return xs[len(xs)//2]
gb = 12.0 * n * k / 1e9
out = {"bytes_GB": round(gb, 3)}
only = os.environ.get("ELT")
for name, fn in (("mu_update_h", lambda: ops.mu_update_h(H, S, G, 1.19e-7, False)),
                 ("mu_update_w", lambda: ops.mu_update_w(W, SW, G, 1.19e-7)),
                 ("kl_update_h", lambda: ops.kl_update_h(H, S, x, 1.19e-7, False)),
                 ("kl_update_w", lambda: ops.kl_update_w(W, SW, x, 1.19e-7))):
    if only and name != only:
        continue
    try:
        ms = t(fn)
    except Exception as e:  # a variant the build does not carry
        out[name] = str(e)[:80]; continue
    out[name] = {"ms": round(ms, 3), "GBs": round(gb / ms * 1e3, 1), "frac_hbm_8TBs": round(gb / ms * 1e3 / 8000, 3)}
if only:
    print(json.dumps(out)); sys.exit(0)
ms = t(lambda: ops.clamp_min(H, 1.19e-7)); out["clamp_min"] = {"ms": round(ms, 3), "GBs": round(8.0 * n * k / 1e9 / ms * 1e3, 1)}
ms = t(lambda: ops.sqnorm(H)); out["sqnorm"] = {"ms": round(ms, 3), "GBs": round(4.0 * n * k / 1e9 / ms * 1e3, 1)}
print(json.dumps(out))
