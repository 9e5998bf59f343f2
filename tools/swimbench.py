#!/usr/bin/env python3
"""The reference's own example as a benchmark: examples/dist_pynmfk_2d_Swim.py (KL / MU, k = 14..18, 20 perturbations x 5000
iterations per k, 1024 x 256) on ONE rank, device resident -- the sweep tests/test_gpu_nmfk.py::test_swim_kl_known_answer_on_one_rank
asserts nopt == 16 on.  Prints one JSON line: seconds per sweep, KL steps per second, and the batched step time per k measured on
the whole-fit entry point alone (20 problems, 2000 steps).  Every fit is small enough for the persistent whole-fit kernel
(csrc/dnmf_small.h); `--fit-loop python` times the per-step path it replaces.
usage: swimbench.py [--itr 5000] [--fit-loop native|python] [--reps 3]"""
import argparse
import contextlib
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pydnmfk_amd.dist_comm import MPI_comm  # noqa: E402
from pydnmfk_amd.engine import HIP_OPS as ops, stack_alloc  # noqa: E402
from pydnmfk_amd.pyDNMFk import PyNMFk  # noqa: E402
from pydnmfk_amd.utils import parse  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--itr", type=int, default=5000)
    ap.add_argument("--fit-loop", default="native", choices=["native", "python"])
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    A = torch.from_numpy(np.ascontiguousarray(np.load(os.path.join(here, "tests", "golden", "data_swim.npz"))["A"].astype(np.float32))).cuda()
    m, n = A.shape

    def sweep():
        comms = MPI_comm(None, 1, 1)
        q = parse()
        q.size, q.rank, q.comm, q.p_r, q.p_c = 1, 0, comms, 1, 1
        q.row_comm, q.col_comm, q.comm1 = comms.cart_1d_row(), comms.cart_1d_column(), comms.comm
        q.fpath, q.fname, q.ftype = "../data/", "swim", "mat"
        q.start_k, q.end_k, q.sill_thr, q.itr, q.init = 14, 18, 0.6, a.itr, "rand"
        q.noise_var, q.verbose, q.norm, q.method, q.checkpoint = 0.016, False, "kl", "mu", False
        q.prune, q.rng, q.results_path = False, "device", tempfile.mkdtemp(prefix="swimbench_") + "/"
        if a.fit_loop == "python":
            q.fit_loop = "python"
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(sys.stderr):
            nopt = PyNMFk(A, factors=None, params=q).fit()
        torch.cuda.synchronize()
        return nopt, time.perf_counter() - t0

    sweep()                                                                     # warm: module load, LDS attributes, allocations
    runs = [sweep() for _ in range(a.reps)]
    secs = sorted(r[1] for r in runs)[len(runs) // 2]
    steps = 5 * 21 * a.itr                                                      # 20 perturbation fits + the regression fit, per k
    out = {"workload": "NMFk on swim %d x %d: KL/MU, k = 14..18, 20 perturbations, %d iterations (reference examples/dist_pynmfk_2d_Swim.py), one rank"
                       % (m, n, a.itr), "fit_loop": a.fit_loop, "nopt": int(runs[-1][0]), "seconds_per_sweep": secs, "kl_steps_per_sec": steps / secs,
           "fits_per_sec": 105 / secs, "batched_step_us": {}}
    if a.fit_loop == "native":
        B, itr = 20, 2000
        for k in (14, 16, 17, 18):
            St = stack_alloc(B, m, n, torch.float32, A.device); St.copy_(A[None] * (1 + 0.016 * torch.rand(B, m, n, device=A.device)))
            W = stack_alloc(B, m, k, torch.float32, A.device); H = stack_alloc(B, k, n, torch.float32, A.device)
            best = None
            for _ in range(3):
                W.copy_(torch.rand(B, m, k, device=A.device)); H.copy_(torch.rand(B, k, n, device=A.device))
                torch.cuda.synchronize(); t0 = time.perf_counter()
                ops.fit("mu", "kl", St, W, H, 1.1920929e-07, True, itr)
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            us = best / itr * 1e6
            kp = 16 if k <= 16 else 32
            out["batched_step_us"][str(k)] = {"us": round(us, 2), "tflops_padded_k%d" % kp: round(B * 8.0 * m * n * kp / us / 1e6, 1),
                                              "tflops_algorithmic": round(B * 8.0 * m * n * k / us / 1e6, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
