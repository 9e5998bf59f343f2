"""NMFk driver on the GPU (real HIP kernels inside every PyNMF.fit) against the statistics captured from the reference's
PyNMFk (tests/golden/nmfk_1x1.npz): same data, seeds, parameters; numpy input -> the reference's numpy RNG stream."""
import numpy as np
import pytest

from tests.conftest import LONG, long_param

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def test_nmfk_on_gpu_matches_reference_statistics(tmp_path, golden_dir):
    from pydnmfk_amd.dist_comm import MPI_comm
    from pydnmfk_amd.pyDNMFk import PyNMFk
    from tests.test_nmfk_cpu import _args, check_against_golden
    z = np.load(golden_dir + "/nmfk_1x1.npz")
    comms = MPI_comm(None, 1, 1)
    nmfk = PyNMFk(z["A"], factors=None, params=_args(tmp_path, comms))
    nopt = nmfk.fit()
    check_against_golden(nmfk, nopt, z)


def test_nmfk_kl_on_gpu_matches_reference_statistics(tmp_path, golden_dir):
    """second reference fixture: five features, KL objective, k = 3..7 (the 16-wide fp32 KL kernels inside every fit)"""
    import json
    from pydnmfk_amd.dist_comm import MPI_comm
    from pydnmfk_amd.pyDNMFk import PyNMFk
    from tests.test_nmfk_cpu import _args, check_against_golden_kl5
    z = np.load(golden_dir + "/nmfk_kl5_1x1.npz")
    comms = MPI_comm(None, 1, 1)
    nmfk = PyNMFk(z["A"], factors=None, params=_args(tmp_path, comms, meta=json.loads(str(z["meta"]))))
    check_against_golden_kl5(nmfk, nmfk.fit(), z)


@pytest.mark.parametrize("fixture,exchange", [("nmfk_hals_1x1.npz", None), ("nmfk_2x1.npz", None), ("nmfk_hals_2x1.npz", None),
                                              ("nmfk_hals_2x1.npz", "native-hosted")])
def test_nmfk_hals_and_two_rank_fixtures_on_gpu(fixture, exchange, golden_dir):
    """BASELINE config 5's method pinned to the REFERENCE: the NMFk sweep with HALS inside every fit, on one
    rank and on a 2 x 1 grid (two processes on the one GPU), against statistics the unmodified reference wrote for the same
    data, seeds and parameters -- and once more with every HALS step sequenced inside the library (dnmf_hals_fro_step_1d over
    the hosted transport)."""
    from tests._mp import run_nmfk_golden
    from tests.test_nmfk_cpu import check_nmfk_fixture
    outs = run_nmfk_golden(fixture, use_hip=True, timeout=600, extra={"exchange": exchange} if exchange else None)
    check_nmfk_fixture(outs, np.load(golden_dir + "/" + fixture), tight=False)


@pytest.mark.parametrize("fixture,world", [("nmfk_hals_1x1.npz", 2), ("nmfk_1x1.npz", 4)])
def test_nmfk_perturbations_over_ranks_on_gpu(fixture, world, golden_dir):
    """`params.nmfk_split = 'perturbations'` with the HIP kernels: `world` processes on the one GPU hold the whole matrix and
    share the perturbation fits (each rank's share runs as ONE batched whole-fit call), every rank clusters the gathered factors.
    The statistics equal the one-rank run's exactly -- a problem's result does not depend on which batch it was fitted in -- and
    meet the reference's 1 x 1 fixture.  (hals_sweep = 'columns': two processes share the GPU here, which the persistent W sweep
    must not be asked to do.)"""
    from tests._mp import run_nmfk_golden
    from tests.test_nmfk_cpu import check_nmfk_fixture
    z = np.load(golden_dir + "/" + fixture)
    extra = {"hals_sweep": "columns"}
    one = run_nmfk_golden(fixture, use_hip=True, timeout=600, extra=dict(extra))
    many = run_nmfk_golden(fixture, use_hip=True, timeout=600, extra=dict(extra, nmfk_split="perturbations"), world=world)
    assert len(many) == world
    for o in many:
        assert o[0] == one[0][0]
        for k in one[0][1]:
            for key, val in one[0][1][k].items():
                np.testing.assert_allclose(np.asarray(o[1][k][key], dtype=np.float64), np.asarray(val, dtype=np.float64), rtol=1e-12, atol=0,
                                           err_msg="k=%d %s" % (k, key))
    if "meta" in z.files:
        check_nmfk_fixture(many, z, tight=False)


def test_nmfk_device_resident_input(tmp_path, golden_dir):
    """CUDA-tensor input: perturbations are drawn on the device (different stream, same distribution) -> the estimate
    and the error levels still match, silhouettes are compared loosely."""
    from pydnmfk_amd.dist_comm import MPI_comm
    from pydnmfk_amd.pyDNMFk import PyNMFk
    from tests.test_nmfk_cpu import _args
    z = np.load(golden_dir + "/nmfk_1x1.npz")
    comms = MPI_comm(None, 1, 1)
    np.random.seed(123)
    nmfk = PyNMFk(torch.from_numpy(z["A"]).cuda(), factors=None, params=_args(tmp_path, comms))
    assert nmfk.fit() == 3
    for k, tol in ((1, 0.05), (2, 0.05), (3, 0.6)):   # k = 3 is at the noise floor: its level depends on the draws
        assert abs(nmfk.stats[k]["avgErr"] / float(z["k%d_avgErr" % k]) - 1) < tol, k
    assert np.min(nmfk.stats[3]["clusterSilhouetteCoefficients"]) > 0.8


def test_cli_end_to_end(tmp_path, golden_dir):
    """main.py (reference main.py:13-88 flags): read a .npy, factorise on the GPU, write the factor layout."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    A = np.load(golden_dir + "/data_swim.npz")["A"].astype(np.float32)
    np.save(tmp_path / "swimcopy.npy", A)
    cmd = [sys.executable, os.path.join(root, "main.py"), "--process=pyDNMF", "--p_r=1", "--p_c=1",
           "--fpath=%s/" % tmp_path, "--fname=swimcopy", "--ftype=npy", "--k=4", "--itr=30", "--norm=fro",
           "--method=mu", "--results_path=%s/res/" % tmp_path]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    err = float(out.stdout.strip().split("relative error =")[-1])
    assert 0.5 < err < 0.7                          # swim k=4: 0.646 after 10, 0.606 after 100 iterations (fixtures)
    W = np.load(tmp_path / "res" / "W_factors" / "W_0.npy")
    H = np.load(tmp_path / "res" / "H_factors" / "H_0.npy")
    assert W.shape == (1024, 4) and H.shape == (4, 256) and W.dtype == np.float32
    assert abs(np.linalg.norm(A - W @ H) / np.linalg.norm(A) - err) < 1e-4


def test_cli_nmfk_end_to_end(tmp_path, golden_dir):
    """main.py --process=pyDNMFk (reference main.py:13-88 flags): rank estimation from the command line on the synthetic
    3-feature problem of the NMFk golden (the reference's own run estimates 3), results layout per k on disk."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    z = np.load(golden_dir + "/nmfk_1x1.npz")
    np.save(tmp_path / "synth.npy", z["A"].astype(np.float32))
    cmd = [sys.executable, os.path.join(root, "main.py"), "--process=pyDNMFk", "--p_r=1", "--p_c=1",
           "--fpath=%s/" % tmp_path, "--fname=synth", "--ftype=npy", "--itr=300", "--norm=fro", "--method=mu",
           "--start_k=1", "--end_k=5", "--perturbations=6", "--noise_var=0.03", "--sill_thr=0.8", "--init=rand",
           "--results_path=%s/res/" % tmp_path]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "Estimated k with NMFk is  3" in out.stdout or "Estimated k with NMFk is 3" in out.stdout, out.stdout[-500:]
    for k in range(1, 6):
        base = tmp_path / "res" / "synth" / str(k)
        assert (base / "W_reg_factors" / "W_0.npy").exists() and (base / "H_reg_factors" / "H_0.npy").exists()


@pytest.mark.parametrize("rng", ["device", "numpy"])
def test_cli_nmfk_device_resident_sweep(tmp_path, rng):
    """main.py --process=pyDNMFk the way BASELINE config 5 runs it (HALS / Frobenius, bf16-stored X, 20 perturbations) on the
    6-feature problem of tests/test_gpu_nmfk_sweep.py: `--rng device` (the CLI default) uploads the block once and draws
    perturbations and the rand init on the GPU, `--rng numpy` is the reference's host stream; both must recover rank 6."""
    import os
    import subprocess
    import sys
    from tests.test_gpu_nmfk_sweep import TRUE_K, synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    np.save(tmp_path / "synth6.npy", synth())
    cmd = [sys.executable, os.path.join(root, "main.py"), "--process=pyDNMFk", "--p_r=1", "--p_c=1",
           "--fpath=%s/" % tmp_path, "--fname=synth6", "--ftype=npy", "--itr=60", "--norm=fro", "--method=hals",
           "--precision=bfloat16", "--start_k=2", "--end_k=10", "--step_k=1", "--perturbations=20", "--noise_var=0.03",
           "--sill_thr=0.8", "--init=rand", "--rng=%s" % rng, "--results_path=%s/res/" % tmp_path]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert ("Estimated k with NMFk is  %d" % TRUE_K) in out.stdout, out.stdout[-500:]


def test_cli_bf16_precision(tmp_path, golden_dir):
    """main.py --precision bfloat16 --method hals (BASELINE config 5 flags): X is held as bf16 on the GPU, the factors
    come back float32 and reproduce the rounded matrix as well as the reported error says."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    A = np.load(golden_dir + "/data_swim.npz")["A"].astype(np.float32)
    np.save(tmp_path / "swimcopy.npy", A)
    cmd = [sys.executable, os.path.join(root, "main.py"), "--process=pyDNMF", "--p_r=1", "--p_c=1",
           "--fpath=%s/" % tmp_path, "--fname=swimcopy", "--ftype=npy", "--k=4", "--itr=30", "--norm=fro",
           "--method=hals", "--precision=bfloat16", "--results_path=%s/res/" % tmp_path]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    err = float(out.stdout.strip().split("relative error =")[-1])
    W = np.load(tmp_path / "res" / "W_factors" / "W_0.npy")
    H = np.load(tmp_path / "res" / "H_factors" / "H_0.npy")
    assert W.shape == (1024, 4) and H.shape == (4, 256) and W.dtype == np.float32
    Ar = torch.from_numpy(A).to(torch.bfloat16).float().numpy()
    assert abs(np.linalg.norm(Ar - W @ H) / np.linalg.norm(Ar) - err) < 1e-4
    assert 0.4 < err < 0.7


def test_runner_front_end(tmp_path, golden_dir):
    """pyDNMFk_Runner (runner.py:12-176): the Runner object is the params bag."""
    from pydnmfk_amd.runner import pyDNMFk_Runner
    A = np.load(golden_dir + "/data_swim.npz")["A"].astype(np.float32)
    np.save(tmp_path / "swimcopy.npy", A)
    with pytest.raises(ValueError):
        pyDNMFk_Runner(process="nope")
    r = pyDNMFk_Runner(itr=30, norm="fro", method="mu", process="pyDNMF")
    res = r.run(grid=[1, 1], fpath=str(tmp_path) + "/", ftype="npy", fname="swimcopy", results_path=str(tmp_path) + "/res/", k=4)
    assert set(res) == {"W", "H", "err"} and res["W"].shape == (1024, 4) and 0.5 < res["err"] < 0.7
    with pytest.raises(ValueError):
        r.run(grid=[1], fpath=str(tmp_path) + "/")


def test_wtsi_known_answer(tmp_path, golden_dir):
    """The reference's end-to-end known answer (examples/dist_pynmfk_1d_wtsi.py:18-44): wtsi.mat (96 x 21), k = 1..8,
    init='nnsvd', MU/FRO, 1000 iterations, sill_thr 0.6, default 20 perturbations -> asserts nopt == 4."""
    from pydnmfk_amd.dist_comm import MPI_comm
    from pydnmfk_amd.pyDNMFk import PyNMFk
    from pydnmfk_amd.utils import parse
    A = np.load(golden_dir + "/data_wtsi.npz")["A"].astype(np.float32)
    comms = MPI_comm(None, 1, 1)
    args = parse()
    args.comm1, args.comm, args.p_r, args.p_c = comms.comm, comms, 1, 1
    args.row_comm, args.col_comm = comms.cart_1d_row(), comms.cart_1d_column()
    args.fpath, args.fname, args.ftype = str(tmp_path) + "/", "wtsi", "mat"
    args.start_k, args.end_k, args.step_k, args.sill_thr = 1, 8, 1, 0.6
    args.itr, args.init, args.verbose, args.norm, args.method = 1000, "nnsvd", False, "fro", "mu"
    args.checkpoint, args.results_path = False, str(tmp_path) + "/results/"
    assert PyNMFk(A, factors=None, params=args).fit() == 4


@LONG       # 28 s; the fp32 known answer above and the bf16 / HALS sweep against the checker (test_gpu_nmfk_sweep.py) stay
def test_wtsi_known_answer_with_bf16_storage(tmp_path, golden_dir):
    """The same known answer with the data held as bfloat16 (params.precision, BASELINE config 5) and HALS: the rank
    estimate of the wtsi example must survive the 8-bit rounding of X (nopt == 4)."""
    from pydnmfk_amd.dist_comm import MPI_comm
    from pydnmfk_amd.pyDNMFk import PyNMFk
    from pydnmfk_amd.utils import parse
    A = np.load(golden_dir + "/data_wtsi.npz")["A"].astype(np.float32)
    for method in ("mu", "hals"):
        comms = MPI_comm(None, 1, 1)
        args = parse()
        args.comm1, args.comm, args.p_r, args.p_c = comms.comm, comms, 1, 1
        args.row_comm, args.col_comm = comms.cart_1d_row(), comms.cart_1d_column()
        args.fpath, args.fname, args.ftype = str(tmp_path) + "/", "wtsi", "mat"
        args.start_k, args.end_k, args.step_k, args.sill_thr = 1, 8, 1, 0.6
        args.itr, args.init, args.verbose, args.norm, args.method = 1000, "nnsvd", False, "fro", method
        args.checkpoint, args.results_path = False, str(tmp_path) + "/results_%s/" % method
        args.precision = "bfloat16"
        assert PyNMFk(A, factors=None, params=args).fit() == 4, method


@pytest.mark.skipif(not __import__("os").environ.get("DNMF_LONG_TESTS"), reason="~15 min: 4 ranks x 105 KL fits x 5000 iterations over the host-staged test transport; set DNMF_LONG_TESTS=1 (log of a run: profiles/r02_swim_2x2_kl_known_answer.log)")
def test_swim_2x2_kl_known_answer():
    """The reference's second end-to-end known answer (examples/dist_pynmfk_2d_Swim.py:23-50): swim.mat on a 2 x 2 grid,
    KL / MU, k = 14..18, 5000 iterations, rand init, noise 0.016, sill_thr 0.6 -> asserts nopt == 16.  Four ranks on the
    one GPU, real HIP kernels, gloo transport."""
    from tests._mp import run_swim_nmfk
    gemm = __import__("os").environ.get("DNMF_LONG_TESTS_GEMM")      # 'bf16x6': the same known answer on the split KL kernels
    outs = run_swim_nmfk((14, 18, 5000) + ((gemm,) if gemm else ()), use_hip=True, timeout=7000)
    assert all(o[0] == 16 for o in outs), outs
    print("swim 2x2 KL known answer (%s): nopt" % (gemm or "fp32"), outs[0][0], "min silhouettes", outs[0][1], "seconds", round(outs[0][2], 1))



def test_swim_kl_known_answer_on_one_rank(tmp_path, golden_dir):
    """The reference's swim known answer (examples/dist_pynmfk_2d_Swim.py:23-50: KL / MU, k = 14..18, 5000 iterations, rand
    init, noise 0.016, sill_thr 0.6, 20 perturbations -> nopt == 16) with every reference parameter EXCEPT the grid: one
    rank, device-resident (`rng = 'device'`), so the 500 000 KL steps (the 16-wide kernels up to k = 16, the 32-wide ones
    beyond) run without a host transport in between and the whole sweep fits the regular GPU tier.  The estimate is a
    statistical property of the factorisations, not of the process grid; the 2 x 2 run of the example itself is
    test_swim_2x2_kl_known_answer (gated: ~15 min over the host-staged test transport)."""
    import time
    from pydnmfk_amd.dist_comm import MPI_comm
    from pydnmfk_amd.pyDNMFk import PyNMFk
    from pydnmfk_amd.utils import parse
    A = torch.from_numpy(np.ascontiguousarray(np.load(golden_dir + "/data_swim.npz")["A"].astype(np.float32))).cuda()
    comms = MPI_comm(None, 1, 1)
    args = parse()
    args.size, args.rank, args.comm, args.p_r, args.p_c = 1, 0, comms, 1, 1
    args.row_comm, args.col_comm, args.comm1 = comms.cart_1d_row(), comms.cart_1d_column(), comms.comm
    args.fpath, args.fname, args.ftype = "../data/", "swim", "mat"
    args.start_k, args.end_k, args.sill_thr, args.itr, args.init = 14, 18, 0.6, 5000, "rand"
    args.noise_var, args.verbose, args.norm, args.method, args.checkpoint = 0.016, False, "kl", "mu", False
    args.prune, args.rng, args.results_path = False, "device", str(tmp_path) + "/"
    t0 = time.time()
    nmfk = PyNMFk(A, factors=None, params=args)
    nopt = nmfk.fit()
    sil = {k: round(float(np.min(v["clusterSilhouetteCoefficients"])), 3) for k, v in nmfk.stats.items()}
    print("swim KL known answer on one rank: nopt", nopt, "min silhouettes", sil, "seconds", round(time.time() - t0, 1))
    assert nopt == 16, (nopt, sil)
    assert sil[16] > 0.6 and sil[17] < 0.6, sil


@pytest.mark.parametrize("gemm", [long_param("fp32"), long_param("bf16x6")])   # (65 s + 31 s: the 2 x 2 KL sweep stays in the default
def test_swim_2x2_kl_nmfk_short(tmp_path, gemm):                               #  tier through ..._through_library_sequenced_steps below)
    """The same example cut to what fits the regular GPU tier (k = 16..17, 800 iterations): every rank of the 2 x 2 grid
    runs the 2D KL choreography inside NMFk, the four ranks agree on the estimate and on the silhouettes, and the
    clustering of k = 17 (one feature too many for the 16 swimmer limbs) is unstable.  The known answer itself needs the
    reference's 5000 iterations (KL / MU converges slowly: at 1000 the k = 16 silhouette is still 0.54 < sill_thr)."""
    from tests._mp import run_swim_nmfk
    # bf16x6: the split KL kernels on every block (8 perturbations instead of 20: the check is the same, the tier stays short)
    outs = run_swim_nmfk((16, 17, 800, gemm) + ((8,) if gemm == "bf16x6" else ()), use_hip=True, timeout=900)
    assert len({o[0] for o in outs}) == 1
    for o in outs[1:]:
        assert o[1] == outs[0][1]
    sil = outs[0][1]
    assert sil[17] < 0.3 and sil[16] > sil[17], sil
    print("swim 2x2 KL short (%s): nopt" % gemm, outs[0][0], "min silhouettes", sil, "seconds", round(outs[0][2], 1))


def test_swim_2x2_kl_nmfk_through_library_sequenced_steps(monkeypatch):
    """The NMFk driver on the 2 x 2 grid with every 2D KL step sequenced inside the library (exchange = 'native-hosted': the C
    steps over the hosted gloo transport).  Pruning is on (the reference's default), so the ranks first agree that the pruned
    slices still follow the partition rule; the sweep must then give the choreography's estimate and silhouettes exactly --
    same kernels, same order, same sums."""
    from tests._mp import run_swim_nmfk
    cfg = (16, 17, 100, "fp32", 3)
    ref = run_swim_nmfk(cfg, use_hip=True, timeout=300)
    monkeypatch.setenv("DNMF_TEST_EXCHANGE", "native-hosted")
    outs = run_swim_nmfk(cfg, use_hip=True, timeout=300)          # (each rank asserts that steps ran inside the library)
    assert [o[0] for o in outs] == [o[0] for o in ref]
    for o, r in zip(outs, ref):
        assert o[1] == r[1], (o[1], r[1])
