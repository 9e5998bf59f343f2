"""Observed deviation of the HIP HALS fits from the reference goldens (single-rank cases): max rel-Frobenius of W, H and |err diff| per case."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests._golden import case_names, load_case, rel_fro
from tests.test_gpu_parity import _args
from pydnmfk_amd.pyDNMF import PyNMF
worst = (0, 0, 0)
for name in case_names():
    if "_1x1_" not in name or "hals" not in name or not name.endswith(("float32", "float32_noW", "float32_prune")): continue
    meta, A, W0, H0, z = load_case(name)
    for itr in meta["itrs"]:
        W, H, err = PyNMF(A, factors=[W0, H0], params=_args(meta["k"], itr, meta["norm"], meta["W_update"], "hals", meta.get("prune", False))).fit()
        dw, dh, de = rel_fro(W, z["r0_fit%d_W" % itr]), rel_fro(H, z["r0_fit%d_H" % itr]), abs(err - float(z["r0_fit%d_err" % itr]))
        print("%-40s itr %4d  dW %.2e dH %.2e derr %.2e" % (name, itr, dw, dh, de))
        worst = (max(worst[0], dw), max(worst[1], dh), max(worst[2], de))
print("worst", worst)
