"""VERDICT r01 #5, measured: one MU/FRO iteration of config 2 (65536 x 4096 fp32, k = 32) walked in row slabs that fit the
256 MiB Infinity Cache -- fused A.H^T + W update on the slab, then W^T.A on the SAME slab while it is still cached -- against
the two full passes.  Same arithmetic (the W rows of a slab are final before its W^T.A), different order.
Needs the tuning build for the 32-row NT workgroups that fill the GPU on a slab:
    python -m pydnmfk_amd.build --tuning
    DNMF_LIB_PATH=tools/_build/libdnmf_hip_tune.so DNMF_NT_KS4=1 DNMF_TN_NT=0 python tools/onepass.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pydnmfk_amd.engine import HIP_OPS as ops, new_gram

m, n, k = 65536, 4096, 32
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
A = torch.rand(m, n, device=dev, generator=g)
W0 = torch.rand(m, k, device=dev, generator=g)
H0 = torch.rand(k, n, device=dev, generator=g)
eps = 1.1920929e-07


def t(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def two_pass(W, H, G, AtW):
    ops.gram_hht(H, G)
    ops.aht_update_w(A, H, G, W, eps)
    ops.gram_wtw(W, G)
    ops.wta(A, W, AtW)
    ops.mu_update_h(H, AtW, G, eps, False)


def slabbed(W, H, G, parts, AtW, rows):
    ops.gram_hht(H, G)
    for i, r0 in enumerate(range(0, m, rows)):
        ops.aht_update_w(A[r0:r0 + rows], H, G, W[r0:r0 + rows], eps)
        ops.wta(A[r0:r0 + rows], W[r0:r0 + rows], parts[i])
    torch.sum(parts, dim=0, out=AtW)
    ops.gram_wtw(W, G)
    ops.mu_update_h(H, AtW, G, eps, False)


def t_graph(fn, reps=30):
    """the same launch sequence replayed from a captured graph: no host time between the launches"""
    fn(); fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=st):
            fn()
    torch.cuda.synchronize()
    for _ in range(3):
        gr.replay()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        gr.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


out = {"m": m, "n": n, "k": k, "ks4": os.environ.get("DNMF_NT_KS4"), "tn_nt": os.environ.get("DNMF_TN_NT")}
G, AtW = new_gram(k, dev), torch.empty(k, n, device=dev)
W, H = W0.clone(), H0.clone()
out["two_pass_ms"] = round(t(lambda: two_pass(W, H, G, AtW)), 4)
try:
    out["two_pass_graph_ms"] = round(t_graph(lambda: two_pass(W, H, G, AtW)), 4)
except Exception as exc:  # noqa: BLE001
    out["graph_error"] = str(exc)[:200]
W1, H1 = W0.clone(), H0.clone()
two_pass(W1, H1, G, AtW)
for rows in (4096, 8192, 16384, 65536):
    parts = torch.empty(m // rows, k, n, device=dev)
    W, H = W0.clone(), H0.clone()
    out["slab_%d_ms" % rows] = round(t(lambda: slabbed(W, H, G, parts, AtW, rows)), 4)
    if "graph_error" not in out:
        try:
            out["slab_%d_graph_ms" % rows] = round(t_graph(lambda: slabbed(W, H, G, parts, AtW, rows)), 4)
        except Exception as exc:  # noqa: BLE001
            out["graph_error"] = str(exc)[:200]
    W2, H2 = W0.clone(), H0.clone()
    slabbed(W2, H2, G, parts, AtW, rows)
    out["slab_%d_dH" % rows] = float((H2 - H1).norm() / H1.norm())
print(json.dumps(out))
