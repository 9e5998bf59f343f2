"""Kernels whose workgroups wait for each other (whole fits of small problems, the persistent HALS W sweep, the one-pass MU/FRO step)
need the GPU to themselves.  When they do not get it the fit must not be lost: PyNMF keeps the initial
factors, switches the process to the launch-chain kernels (dnmf_set_persistent(0)) and fits again -- same update rules
(pyDNMF.py:151-182), results equal to the undisturbed run at the tolerance the two kernel families agree to (fp32 sums in another
association: 2e-3 rel-Frobenius on the factors of these 200..300-step fits, 1e-4 on recon_err)."""
import warnings

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _rel(x, r):
    x, r = np.asarray(x, dtype=np.float64), np.asarray(r, dtype=np.float64)
    return float(np.linalg.norm(x - r) / np.linalg.norm(r))


def _problem(seed, m=1024, n=256, k=8):
    rs = np.random.RandomState(seed)
    A = np.abs(rs.rand(m, k) @ rs.rand(k, n) + 0.01 * rs.randn(m, n)).astype(np.float32)
    return A, rs.rand(m, k).astype(np.float32), rs.rand(k, n).astype(np.float32)


def _restore(lib):
    import pydnmfk_amd.engine as eng
    lib.dnmf_set_persistent(1)
    lib.dnmf_fit_set_timeout(2.0)
    eng._downgraded = False


@pytest.mark.parametrize("method,norm,k", [("mu", "kl", 8), ("mu", "fro", 20), ("hals", "fro", 8)])
def test_a_fit_whose_persistent_kernel_times_out_is_fitted_again(method, norm, k):
    """A patience of a tenth of a microsecond makes the first barrier anybody waits at give up (the abort path for certain); PyNMF.fit
    still returns the factors of the undisturbed fit, warns once, and the process stays on the launch chain until told otherwise."""
    from pydnmfk_amd._lib import lib
    from pydnmfk_amd.pyDNMF import PyNMF
    from tests.test_gpu_parity import _args
    A, W0, H0 = _problem(3, k=k)
    itr = 200
    try:
        args = _args(k, itr, norm, method=method)
        args.fit_loop = "python"                                          # the step loop: no whole-fit kernel
        args.hals_sweep = "columns"
        lib.dnmf_set_persistent(0)
        Wr, Hr, er = PyNMF(A, factors=[W0, H0], params=args).fit()
        lib.dnmf_set_persistent(1)
        assert lib.dnmf_mu_fit_persistent(1024, 256, k) == 1
        assert lib.dnmf_fit_set_timeout(1e-7) == 0
        with warnings.catch_warnings(record=True) as wlist:
            warnings.simplefilter("always")
            W, H, err = PyNMF(A, factors=[W0, H0], params=_args(k, itr, norm, method=method)).fit()
        assert _rel(W, Wr) < 2e-3 and _rel(H, Hr) < 2e-3 and abs(err - er) < 1e-4
        assert lib.dnmf_set_persistent(1) == 0                            # it did time out and switched the process over ...
        assert any("lost its residency" in str(w.message) for w in wlist)   # ... and said so
    finally:
        _restore(lib)


def _shared_gpu_rank(rank, q, bar):
    try:
        import pydnmfk_amd.engine as eng
        from pydnmfk_amd._lib import lib
        from pydnmfk_amd.pyDNMF import PyNMF
        from tests.test_gpu_parity import _args
        torch.cuda.set_device(0)
        B, k, itr = 20, 16, 300
        probs = [_problem(100 * rank + b, k=k) for b in range(B)]

        def run():
            fits = [PyNMF(A, factors=[W0, H0], params=_args(k, itr, "kl")) for A, W0, H0 in probs]
            return PyNMF.fit_batch(fits)
        lib.dnmf_set_persistent(0)                                        # the reference: launch chains need no residency, alone or not
        ref = run()
        lib.dnmf_set_persistent(1)
        lib.dnmf_fit_set_timeout(0.05)
        bar.wait(timeout=120)
        got = []
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for _ in range(4):                                            # 20 problems x 8 slabs = 160 of 256 CUs per process: both at once cannot be resident
                got.append(run())
        worst = 0.0
        for res in got:
            for (W, H, e), (Wr, Hr, er) in zip(res, ref):
                worst = max(worst, _rel(W, Wr), _rel(H, Hr), 10.0 * abs(e - er))
        q.put((rank, None, worst, int(eng._downgraded)))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc(), None, None))


def test_two_processes_share_the_gpu_and_both_finish():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q, bar = ctx.Queue(), ctx.Barrier(2)
    procs = [ctx.Process(target=_shared_gpu_rank, args=(r, q, bar)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, err, worst, down in res:
        assert err is None, "process %d failed:\n%s" % (rank, err)
        assert worst < 2e-3, (rank, worst, down)


def test_shared_gpu_option_keeps_the_process_on_the_launch_chains():
    """`params.shared_gpu = True` (main.py --shared_gpu True): PyNMF switches the process to the launch-chain kernels before the first
    fit -- no kernel that waits for co-resident workgroups runs, so none can time out; same factors as the persistent kernels give."""
    from pydnmfk_amd._lib import lib
    from pydnmfk_amd.pyDNMF import PyNMF
    from tests.test_gpu_parity import _args
    A, W0, H0 = _problem(5, k=8)
    try:
        Wr, Hr, er = PyNMF(A, factors=[W0, H0], params=_args(8, 100, "kl")).fit()
        assert lib.dnmf_mu_fit_persistent(1024, 256, 8) == 1
        args = _args(8, 100, "kl")
        args.shared_gpu = True
        W, H, err = PyNMF(A, factors=[W0, H0], params=args).fit()
        assert lib.dnmf_mu_fit_persistent(1024, 256, 8) == 0 and lib.dnmf_mu_fro_onepass(8192, 4096, 32) == 0
        assert _rel(W, Wr) < 2e-3 and _rel(H, Hr) < 2e-3 and abs(err - er) < 1e-4
    finally:
        _restore(lib)
