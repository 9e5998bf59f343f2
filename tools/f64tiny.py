"""The reference's own test recipe (tests/test_dist_nmf_1d.py:14-46: 24 x 12 float64 data of exact rank 2, k = 2, 2000 iterations, mu-fro /
mu-kl / hals) through PyNMF.fit on one rank: wall time per fit.  With the tuning build, DNMF_F64_TINY=0 shows the chain of float64
primitives the single-workgroup kernel (csrc/dnmf_f64_tiny.hip) replaced:
    python tools/f64tiny.py
    DNMF_LIB_PATH=tools/_build/libdnmf_hip_tune.so DNMF_F64_TINY=0 python tools/f64tiny.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pydnmfk_amd._lib import lib
from pydnmfk_amd.dist_comm import MPI_comm
from pydnmfk_amd.pyDNMF import PyNMF
from pydnmfk_amd.utils import parse

np.random.seed(100)
m, k, n = 24, 2, 12
A = np.random.rand(m, k) @ np.random.rand(k, n)
comms = MPI_comm(None, 1, 1)
out = {"shape": [m, n, k], "itr": 2000, "tiny_kernel": bool(lib.dnmf_f64_fit_tiny(m, n, k, 0)) and os.environ.get("DNMF_F64_TINY", "1") != "0"}
for mthd, norm in (("mu", "fro"), ("mu", "kl"), ("hals", "fro")):
    ts, err = [], None
    for rep in range(4):
        args = parse()
        args.comm1, args.comm, args.p_r, args.p_c, args.k = comms.comm, comms, 1, 1, k
        args.row_comm, args.col_comm = comms.cart_1d_row(), comms.cart_1d_column()
        args.itr, args.init, args.verbose, args.method, args.norm = 2000, "rand", False, mthd, norm
        np.random.seed(7)
        f = PyNMF(A, factors=None, params=args)
        torch.cuda.synchronize()
        t0 = time.time()
        W, H, err = f.fit()
        torch.cuda.synchronize()
        ts.append(time.time() - t0)
    out["%s_%s" % (mthd, norm)] = {"seconds_per_fit": round(min(ts[1:]), 5), "rel_error": float(err)}
print(json.dumps(out))
