"""A batch of MU/Frobenius fits as an NMFk sweep issues them (PyNMF.fit_batch: the perturbations of one k, device resident) with the
one-pass step on and off:   python tools/batchbench.py [B m n k itr]      (default 8 x 32768 x 4096, k = 16, 50 iterations)
Prints one JSON line: seconds per batch and iterations per second of both, and whether the factors agree."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pydnmfk_amd._lib import lib
from pydnmfk_amd.dist_comm import MPI_comm
from pydnmfk_amd.pyDNMF import PyNMF
from pydnmfk_amd.utils import parse

B, m, n, k, itr = (int(x) for x in sys.argv[1:6]) if len(sys.argv) >= 6 else (8, 32768, 4096, 16, 50)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
As = [torch.rand(m, n, device=dev, generator=g) for _ in range(B)]
W0 = [torch.rand(m, k, device=dev, generator=g) for _ in range(B)]
H0 = [torch.rand(k, n, device=dev, generator=g) for _ in range(B)]
comms = MPI_comm(None, 1, 1)


def args():
    a = parse()
    a.comm1, a.comm, a.p_r, a.p_c, a.k = comms.comm, comms, 1, 1, k
    a.row_comm, a.col_comm = comms.cart_1d_row(), comms.cart_1d_column()
    a.itr, a.init, a.verbose, a.prune, a.norm, a.method = itr, "rand", False, False, "fro", "mu"
    return a


def run():
    fits = [PyNMF(As[b], factors=[W0[b], H0[b]], params=args()) for b in range(B)]
    torch.cuda.synchronize()
    t0 = time.time()
    res = PyNMF.fit_batch(fits)
    torch.cuda.synchronize()
    return time.time() - t0, res


out = {"batch": B, "m": m, "n": n, "k": k, "itr": itr, "onepass_auto": int(lib.dnmf_mu_fro_onepass(m, n, k))}
res = {}
for mode, name in ((0, "two_pass"), (2, "one_pass")):
    lib.dnmf_set_onepass(mode)
    run()
    t, r = min((run() for _ in range(2)), key=lambda x: x[0])
    res[name] = r
    out[name] = {"seconds_per_batch": round(t, 4), "iterations_per_sec": round(B * itr / t, 1)}
lib.dnmf_set_onepass(1)
out["max_rel_diff_W"] = max(float(((a[0] - b[0]).abs().max() / b[0].abs().max()).item()) for a, b in zip(res["one_pass"], res["two_pass"]))
out["max_abs_diff_err"] = max(abs(a[2] - b[2]) for a, b in zip(res["one_pass"], res["two_pass"]))
print(json.dumps(out))
