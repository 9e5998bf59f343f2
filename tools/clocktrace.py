"""Which shader clock does the GPU hold under each kernel of the iteration?  python tools/clocktrace.py [m n k]
A one-wave probe kernel (tools/clockprobe.hip) samples s_memtime against the 100 MHz wall clock on its own stream while the main
stream runs one kind of work back to back; the median slope over the loaded part of the window is the clock held.
Build first: hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o tools/_build/libclockprobe.so tools/clockprobe.hip"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pydnmfk_amd.engine import HIP_OPS as f32, HIP_OPS_BF16X6 as x6, new_gram

m, n, k = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (262144, 8192, 64)
here = os.path.dirname(os.path.abspath(__file__))
probe = ctypes.CDLL(os.path.join(here, "_build", "libclockprobe.so"))
probe.clockprobe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
# ALIAS=1 (tuning build + DNMF_ALLOW_ALIAS=1): every row of A is the same row -> A is cache resident, no HBM traffic, same instructions
A = torch.rand(m, n, device=dev, generator=g) if not os.environ.get("ALIAS") else torch.rand(1, n, device=dev, generator=g).expand(m, n)
W = torch.rand(m, k, device=dev, generator=g)
H = torch.rand(k, n, device=dev, generator=g)
AH = torch.empty(m, k, device=dev)
WtA = torch.empty(k, n, device=dev)
G = f32.gram_hht(H, new_gram(k, dev))
side = torch.cuda.Stream()


def trace(name, fn, ms_window):
    """run fn back to back for about ms_window ms under the probe; returns (ms per call, GHz median, GHz min, GHz max)"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    reps = max(3, int(ms_window / e0.elapsed_time(e1)))
    nsamp = int((ms_window * 1.3 + 2.0) * 1000 / 20)            # one sample per ~20 us (5 naps of ~4 us at 2 GHz)
    buf = torch.zeros(2 * nsamp, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    probe.clockprobe_launch(buf.data_ptr(), nsamp, 5, side.cuda_stream)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    b = buf.cpu().view(-1, 2).double()
    dt = b[1:, 1] - b[:-1, 1]
    ghz = (b[1:, 0] - b[:-1, 0]) / dt * 0.1
    t = (b[1:, 1] - b[0, 1]) * 1e-5                              # ms since the probe started
    load = (t > 0.25 * ms * reps) & (t < 0.95 * ms * reps)        # the loaded part of the window (the clock needs ~1 ms to settle)
    x = ghz[load]
    out = {"what": name, "ms": round(ms, 4), "reps": reps, "ghz_median": round(float(x.median()), 3),
           "ghz_p10": round(float(x.quantile(0.1)), 3), "ghz_p90": round(float(x.quantile(0.9)), 3),
           "ghz_idle_tail": round(float(ghz[t > 1.15 * ms * reps].median()), 3) if bool((t > 1.15 * ms * reps).any()) else None}
    print(json.dumps(out), flush=True)
    return out


win = float(os.environ.get("WINDOW_MS", "60"))
Wc = W.clone()
trace("A H^T + W update (fp32 MFMA)", lambda: f32.aht_update_w(A, H, G, Wc, 1e-7), win)
trace("W^T A (fp32 MFMA)", lambda: f32.wta(A, W, WtA), win)
trace("A H^T (fp32 MFMA, plain)", lambda: f32.aht(A, H, AH), win)
trace("||A||^2 (loads only)", lambda: f32.sqnorm(A) if hasattr(f32, "sqnorm") else A.sum(), win)
if os.environ.get("ALIAS"):
    sys.exit(0)
Wc = W.clone()
trace("A H^T + W update (bf16x6)", lambda: x6.aht_update_w(A, H, G, Wc, 1e-7), win)
trace("W^T A (bf16x6)", lambda: x6.wta(A, W, WtA), win)
