"""phase timing (with device syncs) of pynmfk_per_k on the small config-5 shape"""
import os, sys, time, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pydnmfk_amd.dist_comm import MPI_comm
from pydnmfk_amd import pyDNMFk as M, pyDNMF, dist_clustering
from pydnmfk_amd.utils import parse
T = {}
def wrap(obj, name, key):
    fn = getattr(obj, name)
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); T[key] = T.get(key, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, name, w)
wrap(pyDNMF.PyNMF, "fit_batch", "fit_batch"); wrap(pyDNMF.PyNMF, "fit", "fit(single)"); wrap(pyDNMF.PyNMF, "column_err", "column_err")
wrap(dist_clustering.custom_clustering, "fit", "clustering")
orig_sample_fit = M.sample.fit
def sfit(self):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = orig_sample_fit(self); torch.cuda.synchronize(); T["sample"] = T.get("sample", 0.0) + time.perf_counter() - t0; return r
M.sample.fit = sfit
dev = torch.device("cuda"); m, n = 1024, 256
g = torch.Generator(device=dev).manual_seed(7)
x = torch.arange(m, device=dev, dtype=torch.float32)[:, None]
cen = torch.linspace(0.075 * m, m - 0.075 * m, 6, device=dev)[None, :]
Wt = torch.exp(-(x - cen) ** 2 / (2 * (0.044 * m) ** 2))
Ht = torch.rand(6, n, device=dev, generator=g) * (torch.rand(6, n, device=dev, generator=g) < 0.7)
X = (Wt @ Ht + 0.005 * torch.rand(m, n, device=dev, generator=g)).to(torch.bfloat16)
def sweep():
    comms = MPI_comm(None, 1, 1); q = parse()
    q.comm1, q.comm, q.p_r, q.p_c = comms.comm, comms, 1, 1
    q.row_comm, q.col_comm = comms.cart_1d_row(), comms.cart_1d_column(); q.size, q.rank = 1, 0
    q.norm, q.method, q.init, q.itr, q.verbose, q.prune = "fro", "hals", "rand", 100, False, False
    q.start_k, q.end_k, q.step_k, q.fname, q.checkpoint = 2, 16, 1, "c5", False
    q.perturbations, q.noise_var, q.sampling, q.sill_thr = 20, 0.03, "uniform", 0.8
    q.precision, q.results_path, q.timing_stats, q.rng = "bfloat16", tempfile.mkdtemp() + "/", False, "device"
    q.nmfk_batch = True
    torch.cuda.synchronize(); t0 = time.perf_counter(); n_ = M.PyNMFk(X, factors=None, params=q).fit(); torch.cuda.synchronize()
    return n_, time.perf_counter() - t0
import contextlib, io
with contextlib.redirect_stdout(io.StringIO()):
    sweep(); T.clear(); out = sweep()
print("sweep", out, {k: round(v * 1e3, 1) for k, v in T.items()}, "ms; accounted", round(sum(T.values()) * 1e3, 1))
