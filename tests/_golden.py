"""Helpers to load the golden fixtures captured from the reference (tests/golden/make_golden.py)."""
import glob
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def case_names():
    return sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(GOLDEN, "case_*.npz")))


def load_case(name):
    z = np.load(os.path.join(GOLDEN, "case_%s.npz" % name))
    meta = json.loads(str(z["meta"]))
    d = np.load(os.path.join(GOLDEN, "data_%s.npz" % meta["dataset"]))
    dt = np.dtype(meta["dtype"])
    A = np.ascontiguousarray(d["A"].astype(dt))  # swim.mat is Fortran-ordered
    W0 = d["W0"].astype(dt)
    H0 = d["H0"].astype(dt)
    return meta, A, W0, H0, z


def rel_fro(x, ref):
    x = np.asarray(x, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.linalg.norm(x - ref) / max(np.linalg.norm(ref), 1e-300))
