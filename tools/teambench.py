"""One pass over A per MU/Frobenius step (csrc/dnmf_team.h) against the two-pass sequence, same box, same process.
    python tools/teambench.py [m n k]          (default: BASELINE config 2, 65536 x 4096, k = 32)
Prints one JSON line: agreement of W / H after one step (against float64 on the device for both), run-to-run bit identity of the
one-pass step, and ms per step of both sequences (HIP events over `reps` steps, A resident)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pydnmfk_amd._lib import lib
from pydnmfk_amd.engine import HIP_OPS as ops

m, n, k = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (65536, 4096, 32)
reps = int(os.environ.get("REPS", "50"))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(7)
A = torch.rand(m, n, device=dev, generator=g)
W0 = torch.rand(m, k, device=dev, generator=g)
H0 = torch.rand(k, n, device=dev, generator=g)
eps = 1.1920929e-07
out = {"m": m, "n": n, "k": k, "onepass_auto": int(lib.dnmf_mu_fro_onepass(m, n, k))}
lib.dnmf_set_onepass(2)
out["onepass_shape"] = int(lib.dnmf_mu_fro_onepass(m, n, k))


def step(on, W, H, clamp=False):
    lib.dnmf_set_onepass(2 if on else 0)
    ops.mu_fro_step(A, W, H, eps, True, clamp)


def ref64():
    A64, W, H = A.double(), W0.double(), H0.double()
    W = W * (A64 @ H.T) / (W @ (H @ H.T) + eps)
    H = H * (W.T @ A64) / ((W.T @ W) @ H + eps)
    return W, H


W64, H64 = ref64()
res = {}
for on in (0, 1):
    W, H = W0.clone(), H0.clone()
    step(on, W, H)
    torch.cuda.synchronize()
    res[on] = (W, H)
    out["relerr_W_%s" % ("one" if on else "two")] = float(((W.double() - W64).abs().max() / W64.abs().max()).item())
    out["relerr_H_%s" % ("one" if on else "two")] = float(((H.double() - H64).abs().max() / H64.abs().max()).item())
    out["maxrel_W_%s" % ("one" if on else "two")] = float(((W.double() - W64).abs() / (W64.abs() + 1e-30)).max().item())
out["one_vs_two_W"] = float((res[1][0] - res[0][0]).abs().max().item())
out["one_vs_two_H"] = float((res[1][1] - res[0][1]).abs().max().item())
W2, H2 = W0.clone(), H0.clone()
step(1, W2, H2)
torch.cuda.synchronize()
out["bit_identical_rerun"] = bool(torch.equal(W2, res[1][0]) and torch.equal(H2, res[1][1]))
st = torch.zeros(1, dtype=torch.int32)
import ctypes
flag = ctypes.c_int(0)
lib.dnmf_hals_sweep_status(ctypes.byref(flag), None)
out["timed_out"] = int(flag.value)


def timeit(on):
    W, H = W0.clone(), H0.clone()
    for i in range(5):
        step(on, W, H, i % 10 == 0)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(reps):
        step(on, W, H, i % 10 == 0)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for rnd in range(2):
    if not os.environ.get("NOTWO"):
        out["two_pass_ms_%d" % rnd] = round(timeit(0), 4)
    out["one_pass_ms_%d" % rnd] = round(timeit(1), 4)
lib.dnmf_hals_sweep_status(ctypes.byref(flag), None)
out["timed_out_after_timing"] = int(flag.value)
lib.dnmf_set_onepass(1)
print(json.dumps(out))
