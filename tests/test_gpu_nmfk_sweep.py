"""BASELINE config 5 as a parity test: an NMFk sweep k = 2..16 with 20 perturbations, method = 'hals', the data held
as bfloat16 (`params.precision`), on synthetic data with a known number of latent features (2048 x 512, 6 Gaussian
features + noise).  Every one of the 15 x 20 `PyNMF.fit` calls runs on the HIP kernels (the 16-wide kernels for
k <= 16, the HALS column sweeps, the *_bf16a entry points) and is compared with the same sweep driven through the
checker back end (tests/_ops_double.py: the oracle's numpy arithmetic on float(bf16(A))) -- numpy input makes both
sweeps consume the same numpy RNG stream (perturbation seeds 1000 p, then the rand init), so they differ only by fp32
summation order inside the fits.

Budgets: reconstruction error of every fit <= 1e-5 absolute, average error per k <= 1e-5; factors of every fit
rel-Frobenius <= 2e-3 for k up to the true rank (beyond it the factorisation is over-parameterised, the extra
features fit noise and HALS trajectories separate: there the errors are still pinned, the factors are not);
recovered rank == 6 from both back ends.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

TRUE_K = 6


def synth(m=2048, n=512, kt=TRUE_K):
    rs = np.random.RandomState(21)
    x = np.arange(m, dtype=np.float64)
    W = np.stack([np.exp(-(x - c) ** 2 / (2 * 90.0 ** 2)) for c in np.linspace(150, m - 150, kt)], axis=1)
    H = rs.rand(kt, n) * (rs.rand(kt, n) < 0.7)
    return (W @ H + 0.005 * rs.rand(m, n)).astype(np.float32)


def _sweep(A, tmp, tag, ops):
    from pydnmfk_amd.dist_comm import MPI_comm
    from pydnmfk_amd.pyDNMFk import PyNMFk
    from pydnmfk_amd.utils import parse

    class Recording(PyNMFk):
        def pynmfk_per_k(self):
            super().pynmfk_per_k()
            self.walls[self.k] = self.Wall.detach().cpu().numpy().copy()

    comms = MPI_comm(None, 1, 1)
    args = parse()
    args.comm1, args.comm, args.p_r, args.p_c = comms.comm, comms, 1, 1
    args.row_comm, args.col_comm = comms.cart_1d_row(), comms.cart_1d_column()
    args.fpath, args.fname, args.ftype = str(tmp) + "/", "synth6", "npy"
    args.start_k, args.end_k, args.step_k, args.sill_thr, args.itr, args.init = 2, 16, 1, 0.8, 60, "rand"
    args.noise_var, args.verbose, args.norm, args.method, args.checkpoint = 0.03, False, "fro", "hals", False
    args.prune, args.perturbations, args.precision = False, 20, "bfloat16"
    args.results_path = str(tmp) + "/results_%s/" % tag
    nmfk = Recording(A, factors=None, params=args, ops=ops)
    nmfk.walls = {}
    return nmfk, nmfk.fit()


def test_nmfk_hals_bf16_sweep_matches_checker(tmp_path):
    from tests._golden import rel_fro
    from tests._ops_double import OracleOps
    import time
    from threadpoolctl import threadpool_limits
    A = synth()
    t0 = time.time()
    hip, nopt_hip = _sweep(A, tmp_path, "hip", None)
    t1 = time.time()
    with threadpool_limits(limits=4):      # tiny GEMMs: an unbounded BLAS pool on a many-core host only adds contention
        ref, nopt_ref = _sweep(A, tmp_path, "ref", OracleOps())
    print("sweep wall time: HIP %.1f s, checker %.1f s" % (t1 - t0, time.time() - t1))
    assert nopt_hip == nopt_ref == TRUE_K
    worst_err, worst_fac = 0.0, 0.0
    for k in range(2, 17):
        sh, sr = hip.stats[k], ref.stats[k]
        assert len(sh["recon_err"]) == len(sr["recon_err"]) == 20
        d = float(np.max(np.abs(np.asarray(sh["recon_err"]) - np.asarray(sr["recon_err"]))))
        worst_err = max(worst_err, d)
        assert d <= 1e-5, (k, d)
        assert abs(sh["avgErr"] - sr["avgErr"]) <= 1e-5, k
        assert abs(sh["AIC"] / sr["AIC"] - 1) <= 1e-4, k
        assert hip.walls[k].shape == ref.walls[k].shape == (A.shape[0], k, 20)
        if k <= TRUE_K:
            for p in range(20):
                f = rel_fro(hip.walls[k][:, :, p], ref.walls[k][:, :, p])
                worst_fac = max(worst_fac, f)
                assert f <= 2e-3, (k, p, f)
        sil_h = float(np.min(sh["clusterSilhouetteCoefficients"]))
        sil_r = float(np.min(sr["clusterSilhouetteCoefficients"]))
        if k == TRUE_K:
            assert sil_h > 0.95 and sil_r > 0.95
            assert abs(float(sh["L_errDist"]) - float(sr["L_errDist"])) <= 1e-5
            assert np.allclose(sh["L_err"], sr["L_err"], atol=1e-4)
        if k > TRUE_K:
            assert sil_h < 0.5 and sil_r < 0.5, (k, sil_h, sil_r)
    print("worst |recon_err diff| = %.2e, worst factor rel diff (k <= %d) = %.2e" % (worst_err, TRUE_K, worst_fac))
