import sys, os, json
sys.path.insert(0, "/root/repo")
import torch
from pydnmfk_amd.engine import HIP_OPS as ops, new_gram
m, n, k = int(sys.argv[1]) if len(sys.argv) > 1 else 32768, 8192, int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
A = torch.rand(m, n, device=dev, generator=g); W = torch.rand(m, k, device=dev, generator=g); H = torch.rand(k, n, device=dev, generator=g)
G = ops.gram_hht(H, new_gram(k, dev)); Wt = W.clone()
V = torch.empty(m, k, device=dev); V1 = torch.empty(m, k, device=dev); V2 = torch.empty(m, k, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def t(fn, reps=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    x = sorted(s.elapsed_time(e) for s, e in ev)
    return x[len(x)//2]
def fused(): ops.aht_update_w(A, H, G, Wt, 1.19e-7)
def unfused():
    ops.aht(A, H, V); ops.mu_update_w(Wt, V, G, 1.19e-7)
h = n // 2
A1, A2, H1, H2 = A[:, :h], A[:, h:], H[:, :h], H[:, h:]
def split2():
    cur = torch.cuda.current_stream()
    e0 = torch.cuda.Event(); e0.record(cur)
    s1.wait_event(e0); s2.wait_event(e0)
    with torch.cuda.stream(s1): ops.aht(A1, H1, V1)
    with torch.cuda.stream(s2): ops.aht(A2, H2, V2)
    e1, e2 = torch.cuda.Event(), torch.cuda.Event()
    e1.record(s1); e2.record(s2)
    cur.wait_event(e1); cur.wait_event(e2)
    V1.add_(V2)
    ops.mu_update_w(Wt, V1, G, 1.19e-7)
print(json.dumps({"m": m, "k": k, "fused_ms": t(fused), "unfused_ms": t(unfused), "split2_streams_ms": t(split2), "aht_only_ms": t(lambda: ops.aht(A, H, V)), "half_aht_ms": t(lambda: ops.aht(A1, H1, V1))}))
