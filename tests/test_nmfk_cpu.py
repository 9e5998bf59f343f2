"""NMFk driver (pydnmfk_amd.pyDNMFk) on CPU with the checker back end, against statistics captured from the reference's
PyNMFk on the same data, seeds and parameters (tests/golden/make_golden_nmfk.py -> nmfk_1x1.npz).  numpy input makes
`sample` and the rand init consume the reference's exact numpy RNG stream, so the runs differ only by fp32 summation
order inside NMF (300 iterations): silhouettes agree to a few 1e-2, errors to 1e-3 relative, the estimate exactly."""
import json

import numpy as np
import pytest


def _args(tmp, comms, ops=None, meta=None):
    """the parameters of tests/golden/make_golden_nmfk.py (`meta`: the fixture's own record of them, for the second case)"""
    from pydnmfk_amd.utils import parse
    args = parse()
    args.comm1, args.comm, args.p_r, args.p_c = comms.comm, comms, comms.p_r, comms.p_c
    args.row_comm, args.col_comm = comms.cart_1d_row(), comms.cart_1d_column()
    args.fpath, args.fname, args.ftype = str(tmp) + "/", "synth", "npy"
    args.start_k, args.end_k, args.step_k = 1, 5, 1
    args.sill_thr, args.itr, args.init, args.verbose = 0.8, 300, "rand", False
    args.norm, args.method, args.prune = "fro", "mu", False
    args.perturbations, args.noise_var, args.checkpoint = 6, 0.03, False
    args.results_path = str(tmp) + "/results/"
    if meta is not None:
        args.start_k, args.end_k, args.itr, args.norm = meta["start_k"], meta["end_k"], meta["itr"], meta["norm"]
        assert (meta["perturbations"], meta["noise_var"], meta["sill_thr"], meta["method"]) == (6, 0.03, 0.8, "mu")
    return args


def check_against_golden(nmfk, nopt, z, sil_tol=0.08):
    assert nopt == int(z["nopt"]) == 3
    for k in range(1, 6):
        st = nmfk.stats[k]
        ref_sil = z["k%d_clusterSilhouetteCoefficients" % k]
        sil = np.asarray(st["clusterSilhouetteCoefficients"])
        assert sil.shape == ref_sil.shape == (k,)
        if ref_sil.min() > 0.5:       # stable clusterings are reproducible; unstable ones (k > true k) are chaotic
            assert np.allclose(sil, ref_sil, atol=sil_tol), (k, sil, ref_sil)
        else:
            assert sil.min() < 0.6, (k, sil)
        assert abs(st["avgErr"] / float(z["k%d_avgErr" % k]) - 1) < 5e-3, k
        if k <= 3:    # with the reference's own stacking of the H matrices (hall_layout = 'reference') the regression fit
            # starts from the reference's initial H: its error statistics are reproduced, not just approximated
            assert abs(float(st["L_errDist"]) / float(z["k%d_L_errDist" % k]) - 1) < 2e-4, k
            assert np.allclose(st["L_err"], z["k%d_L_err" % k], rtol=2e-3, atol=1e-5), k
        assert np.asarray(st["L_err"]).shape == (40,)
        assert abs(st["AIC"] / float(z["k%d_AIC" % k]) - 1) < 1e-3


def test_nmfk_matches_reference_statistics(tmp_path, golden_dir):
    from pydnmfk_amd.data_io import read_cluster_results
    from pydnmfk_amd.dist_comm import MPI_comm
    from pydnmfk_amd.pyDNMFk import PyNMFk
    from tests._ops_double import OracleOps
    z = np.load(golden_dir + "/nmfk_1x1.npz")
    A = z["A"]
    comms = MPI_comm(None, 1, 1)
    args = _args(tmp_path, comms)
    nmfk = PyNMFk(A, factors=None, params=args, ops=OracleOps())
    nopt = nmfk.fit()
    check_against_golden(nmfk, nopt, z)
    # on-disk layout (data_io.py:175-209): per-k folder with regressed factors and the statistics container
    for k in range(1, 6):
        base = tmp_path / "results" / "synth" / str(k)
        assert (base / "W_reg_factors" / "W_0.npy").exists() and (base / "H_reg_factors" / "H_0.npy").exists()
        st = read_cluster_results(str(base) + "/")
        assert set(st) == {"clusterSilhouetteCoefficients", "avgSilhouetteCoefficients", "L_err", "L_errDist", "avgErr",
                           "ErrTol", "AIC"}
        assert st["ErrTol"].shape == (6,)


def check_against_golden_kl5(nmfk, nopt, z, tight=False):
    """second fixture (nmfk_kl5_1x1.npz): five features, KL objective, k = 3..7 -- the reference estimates 5 (minimum
    silhouette 0.99 at k = 5, 0.72 / 0.33 below it, negative above it).  tight: the float64-accumulating checker back end
    follows the reference's float32 numpy run to 1e-7 on every statistic, unstable clusterings included; the fp32 MFMA
    kernels are held to the levels."""
    assert nopt == int(z["nopt"]) == 5
    for k in range(3, 8):
        st = nmfk.stats[k]
        ref_sil = z["k%d_clusterSilhouetteCoefficients" % k]
        sil = np.asarray(st["clusterSilhouetteCoefficients"])
        assert sil.shape == ref_sil.shape == (k,)
        err, ref_err = st["avgErr"], float(z["k%d_avgErr" % k])
        if tight:
            assert np.allclose(sil, ref_sil, atol=5e-3), (k, sil, ref_sil)
            assert abs(err / ref_err - 1) < 1e-5, (k, err, ref_err)
            assert abs(float(st["L_errDist"]) / float(z["k%d_L_errDist" % k]) - 1) < (1e-5 if k <= 5 else 1e-2), k
            assert abs(st["AIC"] / float(z["k%d_AIC" % k]) - 1) < 1e-6, k
        elif k == 5:
            assert np.allclose(sil, ref_sil, atol=0.08), (k, sil, ref_sil)
            assert abs(err / ref_err - 1) < 2e-2, (k, err, ref_err)
        else:     # over- / under-complete KL fits end in different local minima per summation order: levels only
            assert sil.min() < 0.8, (k, sil)
            assert abs(err / ref_err - 1) < 0.15, (k, err, ref_err)
        assert np.asarray(st["L_err"]).shape == (52,)
    assert abs(float(nmfk.stats[5]["L_errDist"]) / float(z["k5_L_errDist"]) - 1) < 2e-2
    assert abs(nmfk.stats[5]["AIC"] / float(z["k5_AIC"]) - 1) < 1e-2


def test_nmfk_kl_matches_reference_statistics(tmp_path, golden_dir):
    """The KL objective through the NMFk driver against the reference's PyNMFk on a five-feature problem
    (make_golden_nmfk.py kl5; pyDNMFk.py:26-67, 218-258 with dist_nmf.py:806-849 inside every fit)."""
    from pydnmfk_amd.dist_comm import MPI_comm
    from pydnmfk_amd.pyDNMFk import PyNMFk
    from tests._ops_double import OracleOps
    z = np.load(golden_dir + "/nmfk_kl5_1x1.npz")
    comms = MPI_comm(None, 1, 1)
    nmfk = PyNMFk(z["A"], factors=None, params=_args(tmp_path, comms, meta=json.loads(str(z["meta"]))), ops=OracleOps())
    check_against_golden_kl5(nmfk, nmfk.fit(), z, tight=True)


def check_nmfk_fixture(outs, z, tight):
    """Rank outputs of tests/_mp.run_nmfk_golden against a round-4 fixture (nmfk_2x1 / nmfk_hals_1x1 / nmfk_hals_2x1:
    the 3-feature problem, k = 1..5, 6 perturbations; reference: pyDNMFk.py:169-258 over dist_nmf.py:873-934 / 716-751 and
    the clustering allreduces of dist_clustering.py:84-160).  tight = the float64-accumulating checker back end, which
    follows the reference's fp32 numpy run to 1e-4 on every silhouette -- unstable clusterings included -- and to 1e-6 on
    the errors; the fp32 MFMA kernels are held to the reference's statistics where the clustering is stable (k <= 3) and to
    the levels above it."""
    import json
    meta = json.loads(str(z["meta"]))
    assert [o[0] for o in outs] == [int(z["nopt"])] * len(outs) == [3] * len(outs)
    for o in outs[1:]:                       # every rank ends with the same statistics
        for k in o[1]:
            for key in ("clusterSilhouetteCoefficients", "avgErr", "AIC", "L_errDist"):
                assert np.allclose(o[1][k][key], outs[0][1][k][key], rtol=1e-6, atol=1e-9), (k, key)
    for k in range(meta["start_k"], meta["end_k"] + 1):
        st = outs[0][1][k]
        ref = {key: np.asarray(z["k%d_%s" % (k, key)], dtype=np.float64) for key in
               ("clusterSilhouetteCoefficients", "avgSilhouetteCoefficients", "avgErr", "L_errDist", "L_err", "AIC")}
        sil = np.asarray(st["clusterSilhouetteCoefficients"], dtype=np.float64)
        assert sil.shape == ref["clusterSilhouetteCoefficients"].shape == (k,)
        stable = ref["clusterSilhouetteCoefficients"].min() > 0.5
        if tight:
            assert np.allclose(sil, ref["clusterSilhouetteCoefficients"], atol=5e-4), (k, sil, ref["clusterSilhouetteCoefficients"])
            assert abs(float(st["avgErr"]) / float(ref["avgErr"]) - 1) < 2e-6, k
            assert abs(float(st["AIC"]) / float(ref["AIC"]) - 1) < 1e-6, k
        else:
            if stable:
                assert np.allclose(sil, ref["clusterSilhouetteCoefficients"], atol=0.08), (k, sil, ref["clusterSilhouetteCoefficients"])
            else:
                assert sil.min() < 0.6, (k, sil)
            assert abs(float(st["avgErr"]) / float(ref["avgErr"]) - 1) < 5e-3, k
            assert abs(float(st["AIC"]) / float(ref["AIC"]) - 1) < 1e-3, k
        if k <= 3:      # the regression fit starts from the reference's stacked H (hall_layout = 'reference')
            assert abs(float(st["L_errDist"]) / float(ref["L_errDist"]) - 1) < (1e-5 if tight else 2e-3), k
            assert np.allclose(st["L_err"], ref["L_err"], rtol=1e-3 if tight else 5e-3, atol=1e-5), k


@pytest.mark.parametrize("fixture", ["nmfk_hals_1x1.npz", "nmfk_2x1.npz", "nmfk_hals_2x1.npz"])
def test_nmfk_hals_and_two_rank_fixtures(fixture, golden_dir):
    """BASELINE config 5's method (HALS) and NMFk on more than one rank against the reference itself: fixtures written by the
    unmodified reference on 1 x 1 and 2 x 1 grids (make_golden_nmfk.py multirank), our host code over the checker back end."""
    from tests._mp import run_nmfk_golden
    check_nmfk_fixture(run_nmfk_golden(fixture, use_hip=False, timeout=600), np.load(golden_dir + "/" + fixture), tight=True)


@pytest.mark.parametrize("fixture,world", [("nmfk_hals_1x1.npz", 2), ("nmfk_hals_1x1.npz", 4)])
def test_nmfk_perturbations_over_ranks_equal_the_one_rank_run(fixture, world, golden_dir):
    """`params.nmfk_split = 'perturbations'`: the ranks of the job hold the whole matrix and share the perturbation fits
    (rank r fits p = r, r + N, ...), then every rank clusters the gathered factors.  Same seeds per perturbation => the statistics
    of the one-rank run, EXACTLY (same arithmetic on the same operands; nothing is summed across ranks), on every rank -- and
    therefore the reference's own 1 x 1 statistics (the fixture) at the one-rank tolerance."""
    from tests._mp import run_nmfk_golden
    z = np.load(golden_dir + "/" + fixture)
    one = run_nmfk_golden(fixture, use_hip=False, timeout=600)
    many = run_nmfk_golden(fixture, use_hip=False, timeout=600, extra={"nmfk_split": "perturbations"}, world=world)
    assert len(many) == world
    for o in many:
        assert o[0] == one[0][0]
        for k in one[0][1]:
            for key, val in one[0][1][k].items():
                assert np.array_equal(np.asarray(o[1][k][key]), np.asarray(val)), (k, key)
    check_nmfk_fixture(many, z, tight=True)


def test_sample_follows_reference_stream():
    """pyDNMFk.py:26-49: X * (1 + nv + 2 nv U) with the global numpy RNG seeded per perturbation."""
    from pydnmfk_amd.pyDNMFk import sample
    X = np.arange(12, dtype=np.float32).reshape(3, 4) + 1
    got = sample(X, 0.03, "uniform", seed=3000).fit()
    np.random.seed(3000)
    M = 2 * 0.03 * np.random.random_sample(X.shape).astype(X.dtype) + 0.03
    assert np.array_equal(got, np.multiply(X, M + 1))
    assert got.dtype == np.float32 and (got >= X * 1.03 - 1e-6).all() and (got <= X * 1.09 + 1e-6).all()
    p = sample(X, 0.03, "poisson", seed=1).fit()
    np.random.seed(1)
    assert np.array_equal(p, np.random.poisson(X).astype(X.dtype))


def test_checkpoint_roundtrip(tmp_path):
    """utils.py:486-536: the pickle holds a `parse` bag with flag / perturbation / k, named as the reference names it."""
    import pickle
    from pydnmfk_amd.utils import Checkpoint, _BagUnpickler, parse
    params = parse()
    params.results_path, params.rank = str(tmp_path) + "/", 0
    cp = Checkpoint(True, params)
    cp._save_checkpoint(3, 19, 7)
    raw = open(tmp_path / "checkpoint.p", "rb").read()
    assert b"pyDNMFk.utils" in raw and b"pydnmfk_amd" not in raw       # loadable by lanl/pyDNMFk (verified by the generator)
    bag = _BagUnpickler(open(tmp_path / "checkpoint.p", "rb")).load()
    assert (bag.flag, bag.perturbation, bag.k) == (3, 19, 7) and type(bag) is parse
    cp2 = Checkpoint(True, params)
    cp2.load_from_checkpoint()
    assert (cp2.flag, cp2.perturbation, cp2.k) == (3, 19, 7)
    # anything but the bag class is refused
    evil = pickle.dumps(print)
    (tmp_path / "bad").mkdir()
    open(tmp_path / "bad" / "checkpoint.p", "wb").write(evil)
    params.results_path = str(tmp_path / "bad") + "/"
    with pytest.raises(pickle.UnpicklingError):
        Checkpoint(True, params).load_from_checkpoint()


def test_reference_written_checkpoint_loads(tmp_path, golden_dir):
    """tests/golden/ref_checkpoint.p was written by the reference's own Checkpoint._save_checkpoint(2, 11, 5)
    (make_golden_nmfk.py): its pickle names pyDNMFk.utils.parse, which must load here."""
    import shutil
    from pydnmfk_amd.utils import Checkpoint, parse
    shutil.copy(golden_dir + "/ref_checkpoint.p", tmp_path / "checkpoint.p")
    params = parse()
    params.results_path, params.rank = str(tmp_path) + "/", 0
    cp = Checkpoint(True, params)
    cp.load_from_checkpoint()
    assert (cp.flag, cp.perturbation, cp.k) == (2, 11, 5)


def test_resume_at_k_from_a_reference_checkpoint(tmp_path, golden_dir):
    """pyDNMFk.py:188-196: with a checkpoint for (k = 4, flag = 2) in the results folder a new PyNMFk run starts at
    k = 4; the statistics of k < 4 come from the per-k result files of the interrupted run, and the estimate equals
    the uninterrupted run's."""
    import shutil
    from pydnmfk_amd.dist_comm import MPI_comm
    from pydnmfk_amd.pyDNMFk import PyNMFk
    from tests._ops_double import OracleOps
    z = np.load(golden_dir + "/nmfk_1x1.npz")
    comms = MPI_comm(None, 1, 1)
    args = _args(tmp_path, comms)
    args.checkpoint = True
    full = PyNMFk(z["A"], factors=None, params=args, ops=OracleOps())
    nopt = full.fit()
    assert nopt == 3
    res = tmp_path / "results" / "synth"
    assert (res / "checkpoint.p").exists()
    # the interrupted state: k = 4 clustered but not saved, written in the reference's own pickle format
    ref_cp = open(golden_dir + "/ref_checkpoint.p", "rb").read()
    ref_cp = ref_cp.replace(b"K\x05ub.", b"K\x04ub.")                   # k = 5 -> 4 in the fixture's last field
    open(res / "checkpoint.p", "wb").write(ref_cp)
    shutil.rmtree(res / "4"); shutil.rmtree(res / "5")
    args2 = _args(tmp_path, comms)
    args2.checkpoint = True
    again = PyNMFk(z["A"], factors=None, params=args2, ops=OracleOps())
    assert again.fit() == nopt
    assert sorted(again.stats) == [4, 5]                                  # only the unfinished k's were recomputed
    for k in (4, 5):
        assert abs(again.stats[k]["avgErr"] - full.stats[k]["avgErr"]) < 1e-6
