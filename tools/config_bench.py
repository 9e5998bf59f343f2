"""Secondary BASELINE configs on ONE GPU (per-rank proxies of the multi-GPU ones): python tools/config_bench.py [which...]
  c2   : config 2  -- synthetic fp32 X 65536 x 4096, k = 32, MU/FRO, 1 GPU                  -> it/s
  c4   : config 4  -- per-rank block of 131072 x 65536 on a 4 x 2 grid = 32768 x 32768, k = 128, MU/KL step (no exchange)
  c5   : config 5  -- NMFk sweep k = 2..16 step 2, HALS/FRO, perturbations P, itr I on 65536 x 4096: fp32 vs bf16 storage
  c5cli: the same sweep entered through main.py (the reference's entry point): .npy on disk -> data_read -> one upload ->
         PyNMFk with --rng device (and --rng numpy, the host random stream, for comparison)
Prints one JSON line per config."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pydnmfk_amd.dist_comm import MPI_comm
from pydnmfk_amd.dist_nmf import nmf_algorithms_1D
from pydnmfk_amd.utils import parse

dev = torch.device("cuda", 0)
which = sys.argv[1:] or ["c2", "c4", "c5"]


def params(k, norm, method, m, n):
    comms = MPI_comm(None, 1, 1)
    p = parse()
    p.comm1, p.comm, p.p_r, p.p_c, p.k, p.m, p.n = comms.comm, comms, 1, 1, k, m, n
    p.row_comm, p.col_comm = comms.cart_1d_row(), comms.cart_1d_column()
    p.norm, p.method, p.W_update, p.eps = norm, method, True, 1.1920929e-07
    return p


def step_ms(A, k, norm, method, steps=20, warm=3, gemm=None):
    m, n = A.shape
    g = torch.Generator(device=dev); g.manual_seed(2)
    W = torch.rand(m, k, device=dev, generator=g); H = torch.rand(k, n, device=dev, generator=g)
    p = params(k, norm, method, m, n)
    if gemm:
        p.gemm = gemm
    for i in range(warm): nmf_algorithms_1D(A, W, H, params=p).update(clamp=(i % 10 == 0))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): nmf_algorithms_1D(A, W, H, params=p).update(clamp=(i % 10 == 0))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


g = torch.Generator(device=dev); g.manual_seed(1)
if "c2" in which:
    A = torch.rand(65536, 4096, device=dev, generator=g)
    ms = step_ms(A, 32, "fro", "mu", steps=50)
    print(json.dumps({"config": "c2: MU/FRO 65536x4096 fp32 k=32, 1 GPU", "ms_per_iter": round(ms, 4), "iter_per_s": round(1e3 / ms, 1),
                      "algorithmic_hbm_gbs": round(2 * A.numel() * 4 / ms / 1e6, 1)}))
    ms6 = step_ms(A, 32, "fro", "mu", steps=50, gemm="bf16x6")      # k <= 32: no split kernel, forwarded to the fp32 ones
    print(json.dumps({"config": "c2 with params.gemm = 'bf16x6' (k <= 32 runs the fp32 kernels)", "ms_per_iter": round(ms6, 4)}))
    ms16 = step_ms(A.to(torch.bfloat16), 32, "fro", "mu", steps=50)
    print(json.dumps({"config": "c2 with bf16-stored X", "ms_per_iter": round(ms16, 4), "iter_per_s": round(1e3 / ms16, 1)}))
    del A
if "c4" in which:
    A = torch.rand(32768, 32768, device=dev, generator=g)
    ms = step_ms(A, 128, "kl", "mu", steps=10)
    fl = 8.0 * 32768 * 32768 * 128
    print(json.dumps({"config": "c4 per-rank proxy: MU/KL 32768x32768 fp32 k=128 (one block of 131072x65536 on 4x2)", "ms_per_iter": round(ms, 3),
                      "tflops": round(fl / ms / 1e9, 1)}))
    ms6 = step_ms(A, 128, "kl", "mu", steps=10, gemm="bf16x6")
    print(json.dumps({"config": "c4 per-rank proxy with params.gemm = 'bf16x6'", "ms_per_iter": round(ms6, 3), "speedup": round(ms / ms6, 2)}))
    del A
if "c5k" in which:
    A = torch.rand(65536, 4096, device=dev, generator=g)
    Ab = A.to(torch.bfloat16)
    for k in (2, 4, 8, 16, 32):
        print(json.dumps({"config": "c5k: step ms at 65536x4096", "k": k,
                          "mu_f32": round(step_ms(A, k, "fro", "mu", 50), 4), "mu_bf16": round(step_ms(Ab, k, "fro", "mu", 50), 4),
                          "hals_f32": round(step_ms(A, k, "fro", "hals", 50), 4), "hals_bf16": round(step_ms(Ab, k, "fro", "hals", 50), 4)}))
    del A, Ab
if "c5" in which:
    from pydnmfk_amd.pyDNMFk import PyNMFk
    m, n, P, I = 65536, 4096, int(os.environ.get("C5_P", 4)), int(os.environ.get("C5_ITR", 100))
    rs = np.random.RandomState(7)
    Wt = torch.from_numpy(rs.rand(m, 6).astype(np.float32)).to(dev)
    Ht = torch.from_numpy(rs.rand(6, n).astype(np.float32)).to(dev)
    X = Wt @ Ht + 0.01 * torch.rand(m, n, device=dev, generator=g)
    for prec in ("float32", "bfloat16"):
        for method in ("hals", "mu"):
            comms = MPI_comm(None, 1, 1)
            a = parse()
            a.comm1, a.comm, a.p_r, a.p_c = comms.comm, comms, 1, 1
            a.row_comm, a.col_comm = comms.cart_1d_row(), comms.cart_1d_column()
            a.norm, a.method, a.init, a.itr, a.verbose, a.prune = "fro", method, "rand", I, False, False
            a.start_k, a.end_k, a.step_k, a.fname, a.checkpoint = 2, 16, 2, "c5", False
            a.perturbations, a.noise_var, a.sampling, a.sill_thr = P, 0.015, "uniform", 0.6
            a.precision, a.results_path, a.timing_stats = prec, "/tmp/c5_%s_%s/" % (prec, method), False
            Xin = X.to(torch.bfloat16) if prec == "bfloat16" else X
            torch.cuda.synchronize(); t0 = time.perf_counter()
            nopt = PyNMFk(Xin, factors=None, params=a).fit()
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            nfit = P * 8
            print(json.dumps({"config": "c5 proxy: NMFk k=2..16 step 2, %s/FRO, %d perturbations x %d itr, %dx%d, X stored %s" % (
                method.upper(), P, I, m, n, prec), "seconds": round(dt, 2), "nopt": int(nopt),
                "ms_per_iteration_avg": round(dt / (nfit * I) * 1e3, 3)}))

if "c5cli" in which:
    # the path a user reaches from the reference's entry point must be the device-resident one
    import contextlib, io
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import main as cli
    m, n, P, I = 65536, 4096, int(os.environ.get("C5_P", 4)), int(os.environ.get("C5_ITR", 100))
    rs = np.random.RandomState(7)
    X = (rs.rand(m, 6).astype(np.float32) @ rs.rand(6, n).astype(np.float32) + 0.01 * rs.rand(m, n).astype(np.float32))
    os.makedirs("/tmp/c5cli", exist_ok=True)
    np.save("/tmp/c5cli/planted.npy", X)
    del X
    for rng in ("device", "numpy"):
        for prec in ("float32", "bfloat16"):
            argv = ["main.py", "--process=pyDNMFk", "--p_r=1", "--p_c=1", "--fpath=/tmp/c5cli/", "--fname=planted", "--ftype=npy",
                    "--itr=%d" % I, "--norm=fro", "--method=hals", "--precision=%s" % prec, "--start_k=2", "--end_k=16", "--step_k=2",
                    "--perturbations=%d" % P, "--noise_var=0.015", "--sill_thr=0.6", "--init=rand", "--rng=%s" % rng,
                    "--results_path=/tmp/c5cli/res_%s_%s/" % (rng, prec)]
            old = sys.argv
            sys.argv = argv
            buf = io.StringIO()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            with contextlib.redirect_stdout(buf):
                cli.main()
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            sys.argv = old
            est = [ln for ln in buf.getvalue().splitlines() if "Estimated k" in ln]
            print(json.dumps({"config": "c5cli: main.py --process=pyDNMFk k=2..16 step 2, HALS/FRO, %d perturbations x %d itr, %dx%d, X stored %s, --rng %s (file read + upload included)" % (
                P, I, m, n, prec, rng), "seconds": round(dt, 2), "stdout": est[-1] if est else None}))
