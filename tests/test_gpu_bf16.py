"""bfloat16 STORAGE of the data matrix (BASELINE config 5, `--precision bfloat16`): the `*_bf16a` entry points read A as
bf16 from HBM, widen exactly to fp32 in registers and run the same fp32 MFMA arithmetic.  The reference has no bf16
(numpy has none), so the statement checked here is: every bf16a call equals its fp32 twin / the oracle applied to
float(bf16(A)), within the SAME fp32 tolerances as the fp32 tests -- the only approximation is the one-off rounding of A.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

EPS = float(np.finfo(np.float32).eps)
SHAPES = [(512, 384, 64), (300, 260, 32), (257, 131, 33), (130, 72, 128), (24, 12, 2), (1024, 256, 4),
          (97, 53, 7), (640, 200, 16), (2048, 1024, 64), (33, 515, 65), (4099, 1026, 8),
          # rank <= 16 with whole k-tiles of columns: the 16x16x4 kernels (ragged rows, k = 1, k % 4 != 0, a single tile)
          (300, 256, 16), (1000, 512, 5), (129, 128, 1), (4100, 1024, 13), (257, 128, 16), (8192, 2048, 9)]


def _rel(x, ref):
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.linalg.norm(np.asarray(x, dtype=np.float64) - ref) / max(np.linalg.norm(ref), 1e-300))


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from pydnmfk_amd.engine import HIP_OPS
    return HIP_OPS


def _mk(m, n, k, seed=0):
    rs = np.random.RandomState(seed + m + 7 * n + 13 * k)
    A = rs.rand(m, n).astype(np.float32)
    A[rs.rand(m, n) < 0.2] = 0.0
    Ab = torch.from_numpy(A).cuda().to(torch.bfloat16)          # the stored matrix
    A = Ab.float().cpu().numpy()                                # what it means in fp32
    return Ab, A, rs.rand(m, k).astype(np.float32), rs.rand(k, n).astype(np.float32)


def _d(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


@pytest.mark.parametrize("m,n,k", SHAPES)
def test_contractions_and_norms(ops, m, n, k):
    """global_mm(A, H.T) / global_mm(W.T, A) (dist_nmf.py:705) and the two norms of pyDNMF.py:207-217 on bf16-stored A."""
    Ab, A, W, H = _mk(m, n, k)
    A64, W64, H64 = A.astype(np.float64), W.astype(np.float64) * 0.1, H.astype(np.float64)
    AH = ops.aht(Ab, _d(H), torch.full((m, k), 7.0, device="cuda")).cpu().numpy()
    assert _rel(AH, A64 @ H64.T) < 2e-6
    AtW = ops.wta(Ab, _d(W), torch.full((k, n), 7.0, device="cuda")).cpu().numpy()
    assert _rel(AtW, W.T.astype(np.float64) @ A64) < 2e-6
    assert abs(float(ops.sqnorm(Ab)) / float((A64 ** 2).sum()) - 1) < 1e-6
    r = float(ops.resid_sqnorm(Ab, _d((W * 0.1).astype(np.float32)), _d(H)))
    assert abs(r / float(((A64 - (W * 0.1).astype(np.float32).astype(np.float64) @ H64) ** 2).sum()) - 1) < 5e-6


@pytest.mark.parametrize("m,n,k", SHAPES)
def test_bf16a_against_fp32_twin(ops, m, n, k):
    """On float(bf16(A)) the fp32 entry points see the same operand values.  The TN form walks the rows in the same
    order for both storage types -> bitwise equal; the NT form uses 64-wide k-tiles for bf16 (32 for fp32), so only the
    fp32 summation order differs."""
    from pydnmfk_amd.engine import new_gram
    Ab, A, W, H = _mk(m, n, k)
    Af = Ab.float()
    assert torch.equal(ops.wta(Ab, _d(W), torch.empty(k, n, device="cuda")),
                       ops.wta(Af, _d(W), torch.empty(k, n, device="cuda")))
    x = ops.aht(Ab, _d(H), torch.empty(m, k, device="cuda")).cpu().numpy()
    y = ops.aht(Af, _d(H), torch.empty(m, k, device="cuda")).cpu().numpy()
    assert _rel(x, y) < 1e-6
    G = ops.gram_hht(_d(H), new_gram(k, torch.device("cuda")))
    W1, W2 = _d(W), _d(W)
    ops.aht_update_w(Ab, _d(H), G, W1, EPS)
    ops.aht_update_w(Af, _d(H), G, W2, EPS)
    assert _rel(W1.cpu().numpy(), W2.cpu().numpy()) < 1e-6


@pytest.mark.parametrize("m,n,k", SHAPES)
def test_whole_step_matches_oracle(ops, m, n, k):
    """dnmf_mu_fro_step_bf16a vs the oracle's single-rank step (dist_nmf.py:755-771) on float(bf16(A))."""
    from oracle import nmf_oracle as orc
    Ab, A, W, H = _mk(m, n, k)
    for w_update, clamp in ((True, False), (False, False), (True, True)):
        Wd, Hd = _d(W), _d(H)
        ops.mu_fro_step(Ab, Wd, Hd, EPS, w_update, clamp)
        Wr, Hr = W.copy(), H.copy()
        orc.fro_mu_step_local(A, Wr, Hr, np.float32(EPS), W_update=w_update)
        if clamp:
            Wr, Hr = np.maximum(Wr, np.float32(EPS)), np.maximum(Hr, np.float32(EPS))
        assert _rel(Wd.cpu().numpy(), Wr) < 1e-5, (w_update, clamp)
        assert _rel(Hd.cpu().numpy(), Hr) < 1e-5, (w_update, clamp)


def test_strided_and_unaligned_views(ops):
    """Row-strided views and an odd element offset (2-byte aligned only): the generic path must take over."""
    Ab, A, W, H = _mk(200, 131, 16)
    big = torch.zeros(200, 140, dtype=torch.bfloat16, device="cuda")
    view = big[:, 3:134]
    view.copy_(Ab)
    AH = ops.aht(view, _d(H), torch.empty(200, 16, device="cuda")).cpu().numpy()
    assert _rel(AH, A.astype(np.float64) @ H.T.astype(np.float64)) < 2e-6
    AtW = ops.wta(view, _d(W), torch.empty(16, 131, device="cuda")).cpu().numpy()
    assert _rel(AtW, W.T.astype(np.float64) @ A.astype(np.float64)) < 2e-6
    assert abs(float(ops.sqnorm(view)) / float((A.astype(np.float64) ** 2).sum()) - 1) < 1e-6


def _args(k, itr, method, p="bfloat16"):
    from pydnmfk_amd.dist_comm import MPI_comm
    from pydnmfk_amd.utils import parse
    comms = MPI_comm(None, 1, 1)
    args = parse()
    args.comm1, args.comm, args.p_r, args.p_c, args.k = comms.comm, comms, 1, 1, k
    args.row_comm, args.col_comm = comms.cart_1d_row(), comms.cart_1d_column()
    args.itr, args.init, args.verbose, args.prune = itr, "rand", False, False
    args.norm, args.method, args.W_update, args.precision = "fro", method, True, p
    return args


@pytest.mark.parametrize("m,n,k,method,itr", [(4096, 1024, 16, "mu", 20), (3000, 1500, 8, "hals", 10),
                                              (2048, 768, 64, "mu", 12), (1031, 517, 5, "hals", 10)])
def test_fit_with_bf16_precision_matches_oracle(m, n, k, method, itr):
    """PyNMF(..., params.precision='bfloat16').fit() == the oracle's fp32 fit of float(bf16(A)); tolerances are the fp32
    ones of tests/test_gpu_parity.py (mu 1e-4 / 1e-5; hals 2e-3 / 5e-5 relative)."""
    from oracle import nmf_oracle as orc
    from pydnmfk_amd.pyDNMF import PyNMF
    rs = np.random.RandomState(100)
    A = np.abs(rs.rand(m, k) @ rs.rand(k, n) + 0.01 * rs.randn(m, n)).astype(np.float32)
    W0, H0 = rs.rand(m, k).astype(np.float32), rs.rand(k, n).astype(np.float32)
    nmf = PyNMF(A, factors=[W0, H0], params=_args(k, itr, method))
    assert nmf.A_ij.dtype == torch.bfloat16
    W, H, err = nmf.fit()
    Ar = torch.from_numpy(A).to(torch.bfloat16).float().numpy()
    Wr, Hr, err_r = orc.fit_single(Ar, W0, H0, itr, norm="fro", method=method)
    tol_f, tol_e = (2e-3, 5e-5 * max(err_r, 1e-30)) if method == "hals" else (1e-4, 1e-5)
    assert _rel(W, Wr) <= tol_f and _rel(H, Hr) <= tol_f
    assert abs(err - err_r) <= max(tol_e, 1e-5)
    # and the rounding of A itself is a small, known perturbation of the fp32 problem
    _, _, err32 = orc.fit_single(A, W0, H0, itr, norm="fro", method=method)
    assert abs(err - err32) < 5e-3


def test_kl_refuses_bf16():
    from pydnmfk_amd.pyDNMF import PyNMF
    a = _args(4, 2, "mu")
    a.norm = "kl"
    with pytest.raises(TypeError):
        PyNMF(np.ones((8, 8), np.float32), factors=None, params=a)
