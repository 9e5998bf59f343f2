#!/usr/bin/env python3
"""The float64 path (csrc/dnmf_f64.hip) against its roofline: the two big contractions, the KL quotient and whole MU steps through
the choreography (dist_nmf.py over engine.HipOpsF64), HIP-event timed.  fp64 MFMA peak of MI355X: 78.6 TFLOP/s; HBM 8 TB/s.
usage: f64bench.py [m n k]   (default 65536 4096 64)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pydnmfk_amd.dist_comm import MPI_comm  # noqa: E402
from pydnmfk_amd.dist_nmf import nmf_algorithms_1D  # noqa: E402
from pydnmfk_amd.engine import HIP_OPS_F64 as ops  # noqa: E402
from pydnmfk_amd.utils import parse  # noqa: E402

PEAK_TF, PEAK_GBS = 78.6, 8000.0


def timed(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    m, n, k = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (65536, 4096, 64)
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(1)
    A = torch.rand(m, n, dtype=torch.float64, device=dev, generator=g)
    W = torch.rand(m, k, dtype=torch.float64, device=dev, generator=g)
    H = torch.rand(k, n, dtype=torch.float64, device=dev, generator=g)
    AH = torch.empty(m, k, dtype=torch.float64, device=dev)
    AtW = torch.empty(k, n, dtype=torch.float64, device=dev)
    eps = 2.220446049250313e-16
    fl = 2.0 * m * n * k
    out = {"shape": [m, n, k], "peak_tflops_fp64_mfma": PEAK_TF, "kernels": {}}
    for name, fn, flops, byts in (("f64_nt_kernel (A H^T)", lambda: ops.aht(A, H, AH), fl, 8.0 * m * n),
                                  ("f64_tn_kernel + reduce (W^T A)", lambda: ops.wta(A, W, AtW), fl, 8.0 * m * n),
                                  ("f64_nn_rows_kernel<QUOT> (U = A / (W H + eps))", lambda: ops._quot(A, W, H, eps), fl, 16.0 * m * n)):
        ms = timed(fn)
        out["kernels"][name] = {"ms": ms, "tflops": flops / ms / 1e9, "frac_fp64_mfma": flops / ms / 1e9 / PEAK_TF,
                                "algorithmic_gbs": byts / ms / 1e6, "frac_hbm": byts / ms / 1e6 / PEAK_GBS}
    # the two KL products (fused up to k = 64: csrc/dnmf_f64_kl.h), checked against torch on the way
    for name, fn, ref in (("kl_uht ((A / (W H + eps)) H^T)", lambda: ops.kl_uht(A, W, H, eps, AH), lambda: (A / (W @ H + eps)) @ H.t()),
                          ("kl_wtu (W^T (A / (W H + eps)))", lambda: ops.kl_wtu(A, W, H, eps, AtW), lambda: W.t() @ (A / (W @ H + eps)))):
        got, want = fn(), ref()
        err = float((got - want).abs().max() / want.abs().max())
        ms = timed(fn)
        out["kernels"][name] = {"ms": ms, "tflops": 2 * fl / ms / 1e9, "frac_fp64_mfma": 2 * fl / ms / 1e9 / PEAK_TF,
                                "algorithmic_gbs": 8.0 * m * n / ms / 1e6, "rel_err_vs_torch": err}
        del want
    comms = MPI_comm(None, 1, 1)
    for norm in ("fro", "kl"):
        p = parse()
        p.comm1, p.comm, p.p_r, p.p_c, p.k, p.m, p.n = comms.comm, comms, 1, 1, k, m, n
        p.row_comm, p.col_comm = comms.cart_1d_row(), comms.cart_1d_column()
        p.norm, p.method, p.W_update, p.eps = norm, "mu", True, eps
        ms = timed(lambda: nmf_algorithms_1D(A, W, H, params=p).update(), reps=5, warm=2)
        flops = (4.0 if norm == "fro" else 8.0) * m * n * k
        out["mu_%s_step" % norm] = {"ms": ms, "tflops": flops / ms / 1e9, "frac_fp64_mfma": flops / ms / 1e9 / PEAK_TF}
    # the same steps enqueued by ONE library call (dnmf_f64_fit: what PyNMF.fit runs on one rank) -- no Python frame per launch
    for norm in ("fro", "kl"):
        Wf, Hf = W.clone(), H.clone()
        itr = 10
        ms = timed(lambda: ops.fit("mu", norm, A, Wf, Hf, eps, True, itr), reps=3, warm=1) / itr
        flops = (4.0 if norm == "fro" else 8.0) * m * n * k
        out["mu_%s_step_in_fit" % norm] = {"ms": ms, "tflops": flops / ms / 1e9, "frac_fp64_mfma": flops / ms / 1e9 / PEAK_TF,
                                           "note": "per step of a %d-step dnmf_f64_fit call (incl. normalisation and error evaluation once per call)" % itr}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
