"""The one-pass MU/Frobenius step (csrc/dnmf_team.h: teams of workgroups share 16-row slabs of A by columns, exchange their partials
of A H^T inside the kernel and add W_new^T A from the LDS copy) against the checker (oracle/nmf_oracle.py, pinned to the reference's
vectors by tests/test_oracle_golden.py) and against the two-pass sequence it replaces -- dist_nmf.py:716-732 feeding :736-751.

Tolerances (fp32, stated by SURVEY 7 hard part 4): one step from identical state <= 1e-5 relative (largest entry); a 30-step fit
<= 1e-4 rel-Frobenius on W, H and <= 1e-5 on recon_err (against float64 on the checker's factors).  Runs of one shape are bit-identical (fixed summation orders everywhere).
Every test forces the path with dnmf_set_onepass(2) and asserts that the shape takes it -- a silent fall-back to the two passes would
compare that sequence with itself."""
import ctypes

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

EPS = float(np.finfo(np.float32).eps)
# (m, n, k): whole teams, a ragged last slab, rows that leave teams empty, ranks on both sides of the tile sizes, every team size 512 n' allows
SHAPES = [(4096, 2048, 32), (5000, 2048, 17), (4100, 4096, 24), (8192, 2560, 32), (6007, 3072, 31), (16384, 3584, 20),
          # k <= 16: the 16-wide instantiation (half the threads carry the 16 x 16 tile's elements)
          (4096, 2048, 16), (5003, 3072, 9), (8192, 4096, 1), (6000, 2560, 13),
          # n not a multiple of 512: the last member's piece is narrower (down to one lane's four columns)
          (4096, 2052, 32), (5000, 4000, 24), (4100, 3332, 12), (4096, 2300, 16),
          # k <= 16 beyond 4096 columns: teams of up to 16 members
          (4096, 8192, 16), (5000, 6144, 7), (4100, 5000, 12)]


@pytest.fixture(scope="module")
def env():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from pydnmfk_amd._lib import lib
    from pydnmfk_amd.engine import HIP_OPS
    was = lib.dnmf_set_onepass(2)
    yield lib, HIP_OPS
    lib.dnmf_set_onepass(was)


def _mk(m, n, k, seed=0):
    rs = np.random.RandomState(seed + m + 7 * n + 13 * k)
    Ws, Hs = rs.rand(m, k).astype(np.float32), rs.rand(k, n).astype(np.float32)
    A = np.abs(Ws @ Hs + 0.01 * rs.randn(m, n)).astype(np.float32)       # low rank plus noise (SURVEY 8d "parity runs")
    A[rs.rand(m, n) < 0.1] = 0.0                                          # exact zeros, as in swim.mat
    return A, rs.rand(m, k).astype(np.float32), rs.rand(k, n).astype(np.float32)


def _d(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def _maxrel(x, ref):
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.abs(np.asarray(x, dtype=np.float64) - ref).max() / np.abs(ref).max())


@pytest.mark.parametrize("m,n,k", SHAPES)
@pytest.mark.parametrize("clamp", [False, True])
def test_one_step_matches_the_checker_and_the_two_passes(env, m, n, k, clamp):
    from oracle import nmf_oracle as orc
    lib, ops = env
    assert lib.dnmf_mu_fro_onepass(m, n, k) == 1
    A, W0, H0 = _mk(m, n, k)
    Wr, Hr = orc.fro_mu_step_local(A.astype(np.float64), W0.astype(np.float64), H0.astype(np.float64), EPS)
    if clamp:                                                              # pyDNMF.py:155-157
        Wr, Hr = np.maximum(Wr, EPS), np.maximum(Hr, EPS)
    Ad = _d(A)
    W1, H1 = _d(W0), _d(H0)
    ops.mu_fro_step(Ad, W1, H1, EPS, True, clamp)
    assert _maxrel(W1.cpu().numpy(), Wr) < 1e-5 and _maxrel(H1.cpu().numpy(), Hr) < 1e-5
    lib.dnmf_set_onepass(0)
    try:
        W2, H2 = _d(W0), _d(H0)
        ops.mu_fro_step(Ad, W2, H2, EPS, True, clamp)
    finally:
        lib.dnmf_set_onepass(2)
    assert _maxrel(W1.cpu().numpy(), W2.cpu().numpy()) < 1e-5 and _maxrel(H1.cpu().numpy(), H2.cpu().numpy()) < 1e-5
    W3, H3 = _d(W0), _d(H0)                                                # the same bits every time
    ops.mu_fro_step(Ad, W3, H3, EPS, True, clamp)
    assert torch.equal(W1, W3) and torch.equal(H1, H3)
    ops.hals_check()                                                       # no wait of the kernel gave up


def test_padded_views_and_w_fixed(env):
    """rows of A / H with a pitch beyond n, W as a column view of a wider array; W_update = False keeps the two-pass H phase."""
    from oracle import nmf_oracle as orc
    lib, ops = env
    m, n, k = 4096, 2048, 28
    A, W0, H0 = _mk(m, n, k, seed=3)
    Ap = torch.zeros(m, n + 64, device="cuda"); Ap[:, :n] = _d(A)
    Hp = torch.zeros(k, n + 32, device="cuda"); Hp[:, :n] = _d(H0)
    Wp = torch.zeros(m, 40, device="cuda"); Wp[:, :k] = _d(W0)
    Av, Hv, Wv = Ap[:, :n], Hp[:, :n], Wp[:, :k]
    Wr, Hr = orc.fro_mu_step_local(A.astype(np.float64), W0.astype(np.float64), H0.astype(np.float64), EPS)
    ops.mu_fro_step(Av, Wv, Hv, EPS, True, False)
    assert _maxrel(Wv.cpu().numpy(), Wr) < 1e-5 and _maxrel(Hv.cpu().numpy(), Hr) < 1e-5
    assert not Wp[:, k:].any() and not Hp[:, n:].any() and not Ap[:, n:].any()          # nothing written beside the views
    W1, H1 = _d(W0), _d(H0)
    Wf, Hf = orc.fro_mu_step_local(A.astype(np.float64), W0.astype(np.float64), H0.astype(np.float64), EPS, W_update=False)
    ops.mu_fro_step(_d(A), W1, H1, EPS, False, False)
    assert torch.equal(W1, _d(W0)) and _maxrel(H1.cpu().numpy(), Hf) < 1e-5


@pytest.mark.parametrize("m,n,k", [(4096, 2048, 24), (6007, 3072, 31), (5003, 2048, 11)])
def test_fit_through_pynmf_matches_the_checker(env, m, n, k):
    """PyNMF.fit (whole-fit entry point -> mu_fro_step_impl -> the team kernel every step) against oracle.fit_single."""
    from oracle import nmf_oracle as orc
    from pydnmfk_amd.pyDNMF import PyNMF
    from tests.test_gpu_parity import _args
    lib, _ = env
    A, W0, H0 = _mk(m, n, k, seed=5)
    itr = 30
    Wr, Hr, er = orc.fit_single(A, W0, H0, itr, norm="fro", method="mu")
    W, H, err = PyNMF(A, factors=[W0, H0], params=_args(k, itr, "fro")).fit()
    rel = lambda x, r: float(np.linalg.norm(np.asarray(x, dtype=np.float64) - r) / np.linalg.norm(r))
    assert rel(W, Wr) < 1e-4 and rel(H, Hr) < 1e-4
    # recon_err: against float64 on the checker's factors (its own fp32 evaluation of a 1.8e7-term norm is 2e-5 away from that at the larger shape)
    e64 = float(np.linalg.norm(A.astype(np.float64) - Wr.astype(np.float64) @ Hr.astype(np.float64)) / np.linalg.norm(A.astype(np.float64)))
    assert abs(float(err) - e64) < 1e-5 and abs(float(err) - float(er)) < 5e-5


def test_census_that_cannot_complete_leaves_w_untouched(env):
    """A patience of a tenth of a microsecond: the census gives up before every workgroup has been seen (or, on a fast box, a later wait
    does) -- the call ends, the status word says so (PyNMF raises at the end of such a fit), and with the default patience the next
    step is clean.  When it was the census that gave up, W is exactly what it was."""
    from pydnmfk_amd._lib import DnmfError
    lib, ops = env
    m, n, k = 8192, 4096, 32
    A, W0, H0 = _mk(m, n, k, seed=9)
    Ad, W1, H1 = _d(A), _d(W0), _d(H0)
    assert lib.dnmf_fit_set_timeout(1e-7) == 0
    try:
        ops.mu_fro_step(Ad, W1, H1, EPS, True, False)
        torch.cuda.synchronize()
    finally:
        assert lib.dnmf_fit_set_timeout(2.0) == 0
    flag = ctypes.c_int(0)
    assert lib.dnmf_hals_sweep_status(ctypes.byref(flag), None) == 0      # (read and cleared)
    from oracle import nmf_oracle as orc
    Wr, Hr = orc.fro_mu_step_local(A.astype(np.float64), W0.astype(np.float64), H0.astype(np.float64), EPS)
    if not flag.value:                                                     # every workgroup was there within 0.1 us: then the step is right
        assert _maxrel(W1.cpu().numpy(), Wr) < 1e-5 and _maxrel(H1.cpu().numpy(), Hr) < 1e-5
    W2, H2 = _d(W0), _d(H0)
    ops.mu_fro_step(Ad, W2, H2, EPS, True, False)
    torch.cuda.synchronize()
    ops.hals_check()
    assert _maxrel(W2.cpu().numpy(), Wr) < 1e-5 and _maxrel(H2.cpu().numpy(), Hr) < 1e-5


def test_policy_and_switch(env):
    lib, _ = env
    assert lib.dnmf_mu_fro_onepass(65536, 4096, 32) == 1                  # forced by the fixture
    lib.dnmf_set_onepass(1)
    try:
        assert lib.dnmf_mu_fro_onepass(8192, 4096, 32) == 1               # short blocks: measured faster
        assert lib.dnmf_mu_fro_onepass(49152, 4096, 32) == 1 and lib.dnmf_mu_fro_onepass(49152, 2048, 32) == 0
        assert lib.dnmf_mu_fro_onepass(65536, 4096, 32) == 0              # parity with two passes: stays on the launch sequence
        assert lib.dnmf_mu_fro_onepass(65536, 4096, 24) == 1
        assert lib.dnmf_mu_fro_onepass(65536, 4096, 16) == 1 and lib.dnmf_mu_fro_onepass(8192, 4000, 32) == 1
        assert lib.dnmf_mu_fro_onepass(8192, 4002, 32) == 0                # rows of A must be whole 16-byte groups
        assert lib.dnmf_mu_fro_onepass(8192, 8192, 32) == 0               # more than eight 512-column pieces at k > 16
        assert lib.dnmf_mu_fro_onepass(8192, 8192, 16) == 1 and lib.dnmf_mu_fro_onepass(65536, 8192, 16) == 0
    finally:
        lib.dnmf_set_onepass(2)
    lib.dnmf_set_onepass(0)
    try:
        assert lib.dnmf_mu_fro_onepass(8192, 4096, 32) == 0
    finally:
        lib.dnmf_set_onepass(2)


def test_three_hundred_steps_twice_are_bit_identical(env):
    """The exchange hands 16 x k partials between workgroups through tagged granules with no fence: a stale or torn read would change a
    sum somewhere.  300 steps (every slab of every team exchanges 300 times: 1.2 million hand-offs of 4 KiB) run twice from the same
    factors must end in the same bits, and stay finite; the status word stays clear."""
    lib, ops = env
    m, n, k = 16384, 4096, 32
    A, W0, H0 = _mk(m, n, k, seed=21)
    Ad = _d(A)
    ends = []
    for _ in range(2):
        W, H = _d(W0), _d(H0)
        for it in range(300):
            ops.mu_fro_step(Ad, W, H, EPS, True, it % 10 == 0)
        torch.cuda.synchronize()
        ends.append((W, H))
    ops.hals_check()
    assert torch.equal(ends[0][0], ends[1][0]) and torch.equal(ends[0][1], ends[1][1])
    assert bool(torch.isfinite(ends[0][0]).all()) and bool(torch.isfinite(ends[0][1]).all())


@pytest.mark.parametrize("m,n,k", [(4096, 2048, 12), (5000, 2560, 24)])
def test_batched_fits_run_the_one_pass_step_per_problem_and_equal_single_fits(env, m, n, k):
    """A batched whole fit (PyNMF.fit_batch: the perturbations of an NMFk sweep; every other kernel covers the batch with blockIdx.z) runs
    the team kernel once per problem -- it takes the whole device anyway: the factors equal single fits BIT FOR BIT, and the checker."""
    from oracle import nmf_oracle as orc
    from pydnmfk_amd.pyDNMF import PyNMF
    from tests.test_gpu_parity import _args
    lib, _ = env
    assert lib.dnmf_mu_fro_onepass(m, n, k) == 1
    probs = [_mk(m, n, k, seed=31 + b) for b in range(3)]
    itr = 12
    single = [PyNMF(A, factors=[W0, H0], params=_args(k, itr, "fro")).fit() for A, W0, H0 in probs]
    batch = PyNMF.fit_batch([PyNMF(A, factors=[W0, H0], params=_args(k, itr, "fro")) for A, W0, H0 in probs])
    for (W1, H1, e1), (W2, H2, e2) in zip(single, batch):
        assert np.array_equal(W1, W2) and np.array_equal(H1, H2) and abs(e1 - e2) <= 1e-12 * max(1.0, abs(e1))
    Wr, Hr, _ = orc.fit_single(probs[1][0], probs[1][1], probs[1][2], itr, norm="fro", method="mu")
    rel = lambda x, r: float(np.linalg.norm(np.asarray(x, dtype=np.float64) - r) / np.linalg.norm(r))
    assert rel(batch[1][0], Wr) < 1e-4 and rel(batch[1][1], Hr) < 1e-4
