"""KL products, bf16x6 vs fp32 MFMA: error of both against float64 and timings.  python tools/klsplitbench.py m n k [--nocheck]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pydnmfk_amd.engine import HIP_OPS as f32, HIP_OPS_BF16X6 as x6

m, n, k = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (32768, 16384, 16)
check = "--nocheck" not in sys.argv
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
A = torch.rand(m, n, device=dev, generator=g)
W = torch.rand(m, k, device=dev, generator=g)
H = torch.rand(k, n, device=dev, generator=g)
eps = 1.1920929e-07


def t(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    x = sorted(s.elapsed_time(e) for s, e in ev)
    return x[len(x) // 2]


out = {"m": m, "n": n, "k": k}
U0, U1 = torch.empty(m, k, device=dev), torch.empty(m, k, device=dev)
T0, T1 = torch.empty(k, n, device=dev), torch.empty(k, n, device=dev)
f32.kl_uht(A, W, H, eps, U0); x6.kl_uht(A, W, H, eps, U1)
f32.kl_wtu(A, W, H, eps, T0); x6.kl_wtu(A, W, H, eps, T1)
torch.cuda.synchronize()
if check:
    rows = min(m, 2048)
    Ud = A[:rows].double() / (W[:rows].double() @ H.double() + eps)
    ref = Ud @ H.double().t()
    for nm, got in (("f32", U0), ("x6", U1)):
        e = ((got[:rows].double() - ref) / ref).abs()
        out["uht_max_" + nm], out["uht_rms_" + nm] = float(e.max()), float(e.pow(2).mean().sqrt())
    cols = min(n, 1024)
    ref = torch.zeros(k, cols, dtype=torch.float64, device=dev)
    for r0 in range(0, m, 8192):
        Wd = W[r0:r0 + 8192].double()
        ref += Wd.t() @ (A[r0:r0 + 8192, :cols].double() / (Wd @ H[:, :cols].double() + eps))
    for nm, got in (("f32", T0), ("x6", T1)):
        e = ((got[:, :cols].double() - ref) / ref).abs()
        out["wtu_max_" + nm], out["wtu_rms_" + nm] = float(e.max()), float(e.pow(2).mean().sqrt())
    out["uht_x6_vs_f32"] = float(((U1 - U0).abs() / U0.abs()).max())
    out["wtu_x6_vs_f32"] = float(((T1 - T0).abs() / T0.abs()).max())
out["uht_ms_f32"] = round(t(lambda: f32.kl_uht(A, W, H, eps, U0)), 4)
out["uht_ms_x6"] = round(t(lambda: x6.kl_uht(A, W, H, eps, U1)), 4)
out["wtu_ms_f32"] = round(t(lambda: f32.kl_wtu(A, W, H, eps, T0)), 4)
out["wtu_ms_x6"] = round(t(lambda: x6.kl_wtu(A, W, H, eps, T1)), 4)
Wa, Ha = W.clone(), H.clone()
out["step_ms_f32"] = round(t(lambda: f32.mu_kl_step(A, Wa, Ha, eps), reps=10), 4)
Wb, Hb = W.clone(), H.clone()
out["step_ms_x6"] = round(t(lambda: x6.mu_kl_step(A, Wb, Hb, eps), reps=10), 4)
if check:
    Wa, Ha, Wb, Hb = W.clone(), H.clone(), W.clone(), H.clone()
    for i in range(10):
        f32.mu_kl_step(A, Wa, Ha, eps); x6.mu_kl_step(A, Wb, Hb, eps)
    out["W10_x6_vs_f32"] = float((Wa - Wb).norm() / Wa.norm())
    out["H10_x6_vs_f32"] = float((Ha - Hb).norm() / Ha.norm())
print(json.dumps(out))
