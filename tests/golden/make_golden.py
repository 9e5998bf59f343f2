#!/opt/conda/bin/python3.9
"""Generate golden vectors by running the UNMODIFIED reference (lanl/pyDNMFk).

Run in the BUILD container only (the reference never travels to the GPU box):

    OMP_NUM_THREADS=1 /opt/conda/bin/python3.9 tests/golden/make_golden.py

The reference is imported from /root/reference; `mpi4py` (absent from every
interpreter here) is provided by the thread-simulated stand-in under
oracle/mpi_standin (P ranks = P threads, rank-ordered sums).  For every case we
inject identical initial factors through `PyNMF(..., factors=[W0, H0])`
(reference pyDNMF.py:90-96) so no RNG stream has to be matched, and capture

  * `step1`   : one `nmf_algorithms_{1D,2D}.update()` call (dist_nmf.py:66,634)
                from (W0, H0) -- no clamp, no normalisation;
  * `fit<N>`  : `PyNMF.fit()` with itr=N (pyDNMF.py:138-182): W, H per rank
                and the scalar relative error.

Outputs: tests/golden/data_<dataset>.npz (A, W0, H0, global) and
tests/golden/case_<name>.npz (per-rank blocks + index ranges + errors + meta).
Only data (inputs / expected outputs) is written; no reference source is copied.
"""
import json
import os
import sys

os.environ.setdefault("OMP_NUM_THREADS", "1")
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "oracle", "mpi_standin"))
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
from mpi4py import MPI  # noqa: E402  (the stand-in)
from scipy.io import loadmat  # noqa: E402

import pyDNMFk.config as config  # noqa: E402

config.init(0)
from pyDNMFk.dist_comm import MPI_comm  # noqa: E402
from pyDNMFk.dist_nmf import nmf_algorithms_1D, nmf_algorithms_2D  # noqa: E402
from pyDNMFk.pyDNMF import PyNMF  # noqa: E402
from pyDNMFk.utils import determine_block_params, parse  # noqa: E402


# ----------------------------------------------------------------------------- datasets
def ds_t24x12():
    """The reference's own test problem (tests/test_dist_nmf_1d.py:14-20): exact rank 2."""
    np.random.seed(100)
    m, k, n = 24, 2, 12
    W = np.random.rand(m, k)
    H = np.random.rand(k, n)
    return W @ H, k


def ds_r25x13():
    """Ragged split exerciser (utils.py:39-40)."""
    rs = np.random.RandomState(101)
    W = rs.rand(25, 3)
    H = rs.rand(3, 13)
    return W @ H + 0.01 * rs.rand(25, 13), 3


def ds_swim():
    """data/swim.mat: 1024x256 uint8, 65% nnz (config 1 input; exact zeros for KL)."""
    X = loadmat("/root/reference/data/swim.mat")["X"]
    return X, 4


def ds_lowrank(m, n, k, seed=100):
    rs = np.random.RandomState(seed)
    W = rs.rand(m, k)
    H = rs.rand(k, n)
    return np.abs(W @ H + 0.01 * rs.randn(m, n)), k


def ds_t24x12z():
    """t24x12 with two all-zero rows and one all-zero column (exercises utils.py:117-217 pruning)."""
    A, k = ds_t24x12()
    A = A.copy()
    A[[3, 17], :] = 0
    A[:, 5] = 0
    return A, k


def ds_r50x39():
    """Ragged on every non-square grid used below (50 % 4, 39 % 3, 39 % 2 != 0, and the slices of the blocks again)."""
    rs = np.random.RandomState(104)
    W = rs.rand(50, 5)
    H = rs.rand(5, 39)
    return W @ H + 0.01 * rs.rand(50, 39), 5


DATASETS = {
    "r50x39": ds_r50x39,
    "t24x12z": ds_t24x12z,
    "t24x12": ds_t24x12,
    "r25x13": ds_r25x13,
    "swim": ds_swim,
    "lr136x100k32": lambda: ds_lowrank(136, 100, 32),
    "lr200x136k64": lambda: ds_lowrank(200, 136, 64, seed=102),
    "lr150x140k128": lambda: ds_lowrank(150, 140, 128, seed=103),
}


def init_factors(m, n, k, seed):
    rs = np.random.RandomState(seed)
    return rs.rand(m, k), rs.rand(k, n)


# ----------------------------------------------------------------------------- partition (for slicing W0/H0)
def blk(rank, pgrid, shape):
    d = determine_block_params(rank, pgrid, shape)
    s, e = d.determine_block_index_range_asymm()
    return s, e


def factor_slices(rank, p_r, p_c, m, n):
    """Row range of W and column range of H owned by `rank` (pyDNMF.py:83-129, utils.py:97-115)."""
    (rs, cs), (re, ce) = blk(rank, (p_r, p_c), (m, n))
    re, ce = re + 1, ce + 1
    if p_r != 1 and p_c != 1:  # 2d
        i, j = np.unravel_index(rank, (p_r, p_c))
        (ws, _), (we, _) = blk(int(j), (p_c, 1), (re - rs, 1))
        (_, hs), (_, he) = blk(int(i), (1, p_r), (1, ce - cs))
        return (rs + ws, rs + we + 1), (cs + hs, cs + he + 1)
    if p_c == 1:  # W sharded, H replicated
        return (rs, re), (0, n)
    return (0, m), (cs, ce)  # p_r == 1: W replicated, H sharded


# ----------------------------------------------------------------------------- one case
def run_case(name, dataset, grid, norm, dtype, itrs, W_update=True, init_seed=7, method="mu", prune=False):
    A, k = DATASETS[dataset]()
    A = np.asarray(A).astype(dtype)
    m, n = A.shape
    p_r, p_c = grid
    P = p_r * p_c
    W0, H0 = init_factors(m, n, k, init_seed)
    W0, H0 = W0.astype(dtype), H0.astype(dtype)

    def body(rank):
        comm = MPI.COMM_WORLD
        comms = MPI_comm(comm, p_r, p_c)
        out = {}

        def mkargs(itr):
            args = parse()
            args.size, args.rank, args.comm1, args.comm = comm.size, rank, comms.comm, comms
            args.p_r, args.p_c, args.k = p_r, p_c, k
            args.m, args.n = m, n
            args.itr, args.init = itr, "rand"
            args.row_comm, args.col_comm = comms.cart_1d_row(), comms.cart_1d_column()
            args.verbose, args.prune = False, prune
            args.norm, args.method = norm, method
            args.W_update = W_update
            return args

        (rs, cs), (re, ce) = blk(rank, (p_r, p_c), (m, n))
        A_ij = np.ascontiguousarray(A[rs:re + 1, cs:ce + 1])
        (w0, w1), (h0, h1) = factor_slices(rank, p_r, p_c, m, n)
        Wb, Hb = W0[w0:w1].copy(), H0[:, h0:h1].copy()
        out["A_range"] = np.array([rs, re + 1, cs, ce + 1])
        out["W_range"] = np.array([w0, w1])
        out["H_range"] = np.array([h0, h1])

        # --- one bare update() step
        args = mkargs(1)
        nmf = PyNMF(A_ij, factors=[Wb, Hb], params=args)
        if nmf.topo == "2d":
            W1, H1 = nmf_algorithms_2D(nmf.A_ij, nmf.W_ij, nmf.H_ij, params=nmf.params).update()
        else:
            W1, H1 = nmf_algorithms_1D(nmf.A_ij, nmf.W_i, nmf.H_j, params=nmf.params).update()
        out["step1_W"], out["step1_H"] = W1.copy(), H1.copy()
        out["eps"] = np.array(float(nmf.eps))
        out["m_loc_n_loc"] = np.array([nmf.params.m_loc, nmf.params.n_loc])

        # --- full fits
        for itr in itrs:
            args = mkargs(itr)
            Wf, Hf, err = PyNMF(A_ij, factors=[Wb, Hb], params=args).fit()
            out["fit%d_W" % itr], out["fit%d_H" % itr] = np.array(Wf), np.array(Hf)
            out["fit%d_err" % itr] = np.array(float(err))
        return out

    res = MPI.run_ranks(P, body)
    flat = {}
    for r, o in enumerate(res):
        for key, v in o.items():
            flat["r%d_%s" % (r, key)] = v
    meta = dict(name=name, dataset=dataset, grid=list(grid), norm=norm, dtype=np.dtype(dtype).name,
                itrs=list(itrs), W_update=bool(W_update), init_seed=init_seed, k=int(k), m=int(m), n=int(n), method=method, prune=bool(prune),
                generator="reference lanl/pyDNMFk @ /root/reference, python3.9, numpy %s (OpenBLAS, 1 thread), "
                          "mpi4py stand-in with rank-ordered sums" % np.__version__)
    flat["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "case_%s.npz" % name), **flat)
    errs = {itr: float(res[0]["fit%d_err" % itr]) for itr in itrs}
    print("%-34s grid=%s norm=%s %s  err=%s" % (name, grid, norm, np.dtype(dtype).name, errs), flush=True)


def main():
    # datasets (global A, W0, H0 in float64 / native; cases cast)
    for ds, fn in DATASETS.items():
        A, k = fn()
        A = np.asarray(A)
        if ds.startswith("lr"):
            A = A.astype(np.float32)  # cases run float32 only; halves the fixture
        W0, H0 = init_factors(A.shape[0], A.shape[1], k, 7)
        np.savez_compressed(os.path.join(HERE, "data_%s.npz" % ds), A=A, W0=W0, H0=H0, k=np.array(k))

    f32, f64 = np.float32, np.float64
    # (i) the reference's own 24x12 k=2 test shape, every grid, both norms, both dtypes
    for grid in ([1, 1], [2, 1], [1, 2], [2, 2]):
        for norm in ("fro", "kl"):
            for dt in (f32, f64):
                g = "%dx%d" % tuple(grid)
                run_case("t24x12_%s_%s_%s" % (g, norm, np.dtype(dt).name), "t24x12", grid, norm, dt, (1, 10, 100))
    # (ii) ragged splits
    for grid in ([3, 1], [1, 3], [2, 2]):
        for norm in ("fro", "kl"):
            g = "%dx%d" % tuple(grid)
            run_case("r25x13_%s_%s_float32" % (g, norm), "r25x13", grid, norm, f32, (1, 10))
    # (iii) swim (config 1) fp32: 1x1, 4x1 FRO ; 1x1, 2x2 KL (zeros in X)
    run_case("swim_1x1_fro_float32", "swim", [1, 1], "fro", f32, (10, 100))
    run_case("swim_4x1_fro_float32", "swim", [4, 1], "fro", f32, (10, 100))
    run_case("swim_1x4_fro_float32", "swim", [1, 4], "fro", f32, (10,))
    run_case("swim_1x1_kl_float32", "swim", [1, 1], "kl", f32, (10, 100))
    run_case("swim_2x2_kl_float32", "swim", [2, 2], "kl", f32, (10,))
    run_case("swim_2x2_fro_float32", "swim", [2, 2], "fro", f32, (10,))
    # (iv) regression mode
    run_case("swim_1x1_fro_float32_noW", "swim", [1, 1], "fro", f32, (10,), W_update=False)
    run_case("t24x12_2x1_kl_float32_noW", "t24x12", [2, 1], "kl", f32, (10,), W_update=False)
    # (v) MFMA-relevant k on tile-unfriendly shapes
    for ds in ("lr136x100k32", "lr200x136k64", "lr150x140k128"):
        for grid in ([1, 1], [2, 1], [1, 2], [2, 2]):
            for norm in ("fro", "kl"):
                g = "%dx%d" % tuple(grid)
                run_case("%s_%s_%s_float32" % (ds, g, norm), ds, grid, norm, f32, (10,))


def main_hals():
    """HALS / Frobenius cases (dist_nmf.py:411-470, :873-934) -- SURVEY 8f "next" row 1."""
    f32, f64 = np.float32, np.float64
    for grid in ([1, 1], [2, 1], [1, 2], [2, 2]):
        g = "%dx%d" % tuple(grid)
        run_case("t24x12_%s_hals_float32" % g, "t24x12", grid, "fro", f32, (1, 10, 100), method="hals")
    run_case("t24x12_1x1_hals_float64", "t24x12", [1, 1], "fro", f64, (10,), method="hals")
    for grid in ([3, 1], [2, 2]):
        g = "%dx%d" % tuple(grid)
        run_case("r25x13_%s_hals_float32" % g, "r25x13", grid, "fro", f32, (1, 10), method="hals")
    for grid in ([1, 1], [4, 1], [2, 2]):
        g = "%dx%d" % tuple(grid)
        run_case("swim_%s_hals_float32" % g, "swim", grid, "fro", f32, (10,), method="hals")
    run_case("swim_1x1_hals_float32_noW", "swim", [1, 1], "fro", f32, (10,), W_update=False, method="hals")
    for ds in ("lr136x100k32", "lr200x136k64", "lr150x140k128"):
        for grid in ([1, 1], [2, 1], [1, 2]):
            g = "%dx%d" % tuple(grid)
            run_case("%s_%s_hals_float32" % (ds, g), ds, grid, "fro", f32, (10,), method="hals")


def main_prune():
    """Zero-row/column pruning (pyDNMF.py:99-101, utils.py:117-217): factors come back un-pruned (and float64)."""
    A, k = DATASETS["t24x12z"]()
    W0, H0 = init_factors(A.shape[0], A.shape[1], k, 7)
    np.savez_compressed(os.path.join(HERE, "data_t24x12z.npz"), A=A, W0=W0, H0=H0, k=np.array(k))
    for grid in ([1, 1], [2, 1], [1, 2], [2, 2]):
        g = "%dx%d" % tuple(grid)
        run_case("t24x12z_%s_fro_float32_prune" % g, "t24x12z", grid, "fro", np.float32, (10,), prune=True)
    run_case("t24x12z_1x1_kl_float32_prune", "t24x12z", [1, 1], "kl", np.float32, (10,), prune=True)


def main_nonsquare():
    """Non-square 2D grids (BASELINE config 4 is 4x2): the slice rule (utils.py:99-103) and the allgather / Reduce_scatter
    pairing (dist_nmf.py:145-205, :268-343) only distinguish the size-p_r group from the size-p_c group when p_r != p_c."""
    A, k = DATASETS["r50x39"]()
    W0, H0 = init_factors(A.shape[0], A.shape[1], k, 7)
    np.savez_compressed(os.path.join(HERE, "data_r50x39.npz"), A=A, W0=W0, H0=H0, k=np.array(k))
    f32 = np.float32
    for grid in ([4, 2], [2, 3], [3, 2], [2, 4]):
        g = "%dx%d" % tuple(grid)
        for norm, method in (("fro", "mu"), ("kl", "mu"), ("fro", "hals")):
            tag = "hals" if method == "hals" else norm
            run_case("r50x39_%s_%s_float32" % (g, tag), "r50x39", grid, norm, f32, (1, 10), method=method)
    for grid in ([4, 2], [2, 3]):
        g = "%dx%d" % tuple(grid)
        for norm, method in (("fro", "mu"), ("kl", "mu"), ("fro", "hals")):
            tag = "hals" if method == "hals" else norm
            run_case("lr200x136k64_%s_%s_float32" % (g, tag), "lr200x136k64", grid, norm, f32, (10,), method=method)
    run_case("lr150x140k128_4x2_kl_float32", "lr150x140k128", [4, 2], "kl", f32, (10,))     # config 4's rank and grid
    run_case("swim_4x2_kl_float32", "swim", [4, 2], "kl", f32, (10,))                      # zeros in X on 4x2


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "nonsquare":
        main_nonsquare()
    elif len(sys.argv) > 1 and sys.argv[1] == "prune":
        main_prune()
    elif len(sys.argv) > 1 and sys.argv[1] == "hals":
        main_hals()      # adds the HALS cases without touching the MU fixtures
    else:
        main()
        main_hals()
        main_prune()
        main_nonsquare()
