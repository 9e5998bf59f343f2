"""bf16x6 contractions vs the fp32-MFMA ones: accuracy against float64 and timing.  python tools/splitbench.py m n k"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pydnmfk_amd.engine import HIP_OPS as f32, HIP_OPS_BF16X6 as x6, new_gram

m, n, k = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (262144, 8192, 64)
check = os.environ.get("CHECK", "1") != "0" and "--nocheck" not in sys.argv
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
A = torch.rand(m, n, device=dev, generator=g)
if "--bf16" in sys.argv:
    A = A.to(torch.bfloat16)          # bf16-stored X: fp32-MFMA bf16a kernels vs three bf16 piece products
W = torch.rand(m, k, device=dev, generator=g)
H = torch.rand(k, n, device=dev, generator=g)


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
AH0, AH1 = torch.empty(m, k, device=dev), torch.empty(m, k, device=dev)
AtW0, AtW1 = torch.empty(k, n, device=dev), torch.empty(k, n, device=dev)
f32.aht(A, H, AH0); x6.aht(A, H, AH1)
f32.wta(A, W, AtW0); x6.wta(A, W, AtW1)
torch.cuda.synchronize()
if check:
    rows = min(m, 4096)
    ref = (A[:rows].double() @ H.double().t())
    out["aht_relerr_f32"] = float(((AH0[:rows].double() - ref).abs() / ref).max())
    out["aht_relerr_x6"] = float(((AH1[:rows].double() - ref).abs() / ref).max())
    out["aht_rms_f32"] = float(((AH0[:rows].double() - ref) / ref).pow(2).mean().sqrt())
    out["aht_rms_x6"] = float(((AH1[:rows].double() - ref) / ref).pow(2).mean().sqrt())
    cols = min(n, 1024)
    ref = torch.zeros(k, cols, dtype=torch.float64, device=dev)
    for r0 in range(0, m, 32768):
        ref += W[r0:r0 + 32768].double().t() @ A[r0:r0 + 32768, :cols].double()
    out["wta_relerr_f32"] = float(((AtW0[:, :cols].double() - ref).abs() / ref).max())
    out["wta_relerr_x6"] = float(((AtW1[:, :cols].double() - ref).abs() / ref).max())
    out["wta_rms_f32"] = float(((AtW0[:, :cols].double() - ref) / ref).pow(2).mean().sqrt())
    out["wta_rms_x6"] = float(((AtW1[:, :cols].double() - ref) / ref).pow(2).mean().sqrt())
    out["aht_x6_vs_f32"] = float(((AH1 - AH0).abs() / AH0).max())
    out["wta_x6_vs_f32"] = float(((AtW1 - AtW0).abs() / AtW0).max())
G = f32.gram_hht(H, new_gram(k, dev))
out["aht_ms_f32"] = round(t(lambda: f32.aht(A, H, AH0)), 4)
out["aht_ms_x6"] = round(t(lambda: x6.aht(A, H, AH1)), 4)
out["wta_ms_f32"] = round(t(lambda: f32.wta(A, W, AtW0)), 4)
out["wta_ms_x6"] = round(t(lambda: x6.wta(A, W, AtW1)), 4)
Wc = W.clone()
out["ahtupd_ms_f32"] = round(t(lambda: f32.aht_update_w(A, H, G, Wc, 1e-7)), 4)
Wc = W.clone()
out["ahtupd_ms_x6"] = round(t(lambda: x6.aht_update_w(A, H, G, Wc, 1e-7)), 4)
Wa, Ha = W.clone(), H.clone()
out["step_ms_f32"] = round(t(lambda: f32.mu_fro_step(A, Wa, Ha, 1.19e-7), reps=20), 4)
Wb, Hb = W.clone(), H.clone()
out["step_ms_x6"] = round(t(lambda: x6.mu_fro_step(A, Wb, Hb, 1.19e-7), reps=20), 4)
if check:
    Wa, Ha, Wb, Hb = W.clone(), H.clone(), W.clone(), H.clone()
    for i in range(20):
        f32.mu_fro_step(A, Wa, Ha, 1.19e-7); x6.mu_fro_step(A, Wb, Hb, 1.19e-7)
    out["W20_x6_vs_f32"] = float((Wa - Wb).norm() / Wa.norm())
    out["H20_x6_vs_f32"] = float((Ha - Hb).norm() / Ha.norm())
print(json.dumps(out))
