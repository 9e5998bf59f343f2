"""Whole fits with the default arithmetic and with gemm='bf16x6' on awkward data (sparse, wide dynamic range, exactly low rank):
final relative errors and factors side by side."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from pydnmfk_amd.dist_comm import MPI_comm
from pydnmfk_amd.pyDNMF import PyNMF
from pydnmfk_amd.utils import parse

rs = np.random.RandomState(0)
m, n = 4096, 1024


def data(kind, k):
    W, H = rs.rand(m, k), rs.rand(k, n)
    if kind == "lowrank":
        return (W @ H).astype(np.float32)
    if kind == "sparse":
        A = (W @ H) * (rs.rand(m, n) < 0.05)
        return A.astype(np.float32)
    if kind == "wide":
        A = (W @ H) * np.exp(12 * (rs.rand(m, 1) - 0.5)) * np.exp(8 * (rs.rand(1, n) - 0.5))
        return A.astype(np.float32)
    raise ValueError(kind)


bad = 0
for kind in ("lowrank", "sparse", "wide"):
    for norm, k, itr in (("fro", 40, 200), ("kl", 12, 200), ("kl", 64, 100), ("fro", 100, 100)):
        A = data(kind, k)
        W0, H0 = rs.rand(m, k).astype(np.float32), rs.rand(k, n).astype(np.float32)
        res = []
        for gemm in ("fp32", "bf16x6"):
            comms = MPI_comm(None, 1, 1)
            p = parse()
            p.comm1, p.comm, p.p_r, p.p_c, p.k = comms.comm, comms, 1, 1, k
            p.row_comm, p.col_comm = comms.cart_1d_row(), comms.cart_1d_column()
            p.norm, p.method, p.itr, p.init, p.verbose, p.prune, p.W_update, p.gemm = norm, "mu", itr, "rand", False, False, True, gemm
            W, H, err = PyNMF(A, factors=[W0.copy(), H0.copy()], params=p).fit()
            res.append((np.asarray(W), np.asarray(H), err))
        dW = np.linalg.norm(res[0][0] - res[1][0]) / np.linalg.norm(res[0][0])
        dH = np.linalg.norm(res[0][1] - res[1][1]) / np.linalg.norm(res[0][1])
        ok = np.isfinite(res[1][0]).all() and abs(res[0][2] - res[1][2]) <= 1e-4 * max(1e-3, res[0][2]) + 2e-6
        bad += not ok
        print("%-8s %-3s k=%-3d itr=%-3d err fp32 %.6f x6 %.6f  dW %.1e dH %.1e %s" % (kind, norm, k, itr, res[0][2], res[1][2], dW, dH, "" if ok else "<-- CHECK"))
print("failures:", bad)
