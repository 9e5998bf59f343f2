"""End-to-end parity on the GPU through the reference's own entry point, PyNMF(...).fit():
  * against the golden vectors captured from the reference (single-rank cases; multi-rank grids are
    covered by the gloo choreography tests + per-kernel GPU tests, a GPU box has one device);
  * against the oracle on seeded problems too big for fixtures;
  * at BASELINE config-2 size (65536 x 4096, k=32) through size-independent properties.

Tolerances (north_star "within a stated fp32 tolerance"):
  one update step from identical state: rel-Frobenius <= 1e-5
  fits of <= 100 iterations: rel-Frobenius <= 1e-4 on W and H, |recon_err difference| <= 1e-5
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from tests._golden import case_names, load_case, rel_fro  # noqa: E402

SINGLE = [c for c in case_names() if "_1x1_" in c and c.endswith(("float32", "float32_noW", "float32_prune"))]


def _tols(meta):
    """(per-step, fit factors, fit error): HALS sweeps subtract nearly equal terms, so fp32 trajectories are only
    reproducible to ~1e-3 after 10 sweeps (even between two numpy builds: tests/test_oracle_golden.py::test_fit)."""
    if meta.get("method") == "hals":      # (round 4, observed on the GPU over the single-rank HALS goldens -- tools/dbg/hals_dev.py:
        # W <= 1.05e-3 (k = 128, over-parameterised), H <= 1.1e-4, recon_err <= 1.4e-6; regression mode (W fixed, err ~ 30): 3e-5 relative)
        return 5e-5, 2e-3, (1e-5 if meta.get("W_update", True) else 5e-5)
    return 1e-5, 1e-4, 1e-5


def _args(k, itr, norm, W_update=True, method="mu", prune=False):
    from pydnmfk_amd.dist_comm import MPI_comm
    from pydnmfk_amd.utils import parse
    comms = MPI_comm(None, 1, 1)
    args = parse()
    args.comm1, args.comm, args.p_r, args.p_c, args.k = comms.comm, comms, 1, 1, k
    args.row_comm, args.col_comm = comms.cart_1d_row(), comms.cart_1d_column()
    args.itr, args.init, args.verbose, args.prune = itr, "rand", False, prune
    args.norm, args.method, args.W_update = norm, method, W_update
    return args


@pytest.mark.parametrize("fit_loop", ["native", "python"])
@pytest.mark.parametrize("name", SINGLE)
def test_fit_matches_reference_golden(name, fit_loop):
    """Both ways a one-rank fit can run are held to the reference's vectors: the whole-fit entry points (`native`, the default: for the
    small goldens the persistent kernels of csrc/dnmf_small.h) and the per-step loop over nmf_algorithms_1D.update (`python`)."""
    from pydnmfk_amd.pyDNMF import PyNMF
    meta, A, W0, H0, z = load_case(name)
    _, tol_fit, tol_err = _tols(meta)
    for itr in meta["itrs"]:
        args = _args(meta["k"], itr, meta["norm"], meta["W_update"], meta.get("method", "mu"), meta.get("prune", False))
        if fit_loop == "python":
            args.fit_loop = "python"
        W, H, err = PyNMF(A, factors=[W0, H0], params=args).fit()
        # numpy in -> numpy out, in the dtype the reference hands back (float64 after unprune, utils.py:195,198)
        assert isinstance(W, np.ndarray) and W.dtype == z["r0_fit%d_W" % itr].dtype and H.dtype == z["r0_fit%d_H" % itr].dtype
        assert rel_fro(W, z["r0_fit%d_W" % itr]) <= tol_fit, itr
        assert rel_fro(H, z["r0_fit%d_H" % itr]) <= (5e-4 if meta.get("method") == "hals" else tol_fit), itr   # (H carries no column cancellation)
        ref = float(z["r0_fit%d_err" % itr])
        assert abs(err - ref) <= tol_err * max(1.0, abs(ref)), itr


@pytest.mark.parametrize("name", SINGLE)
def test_single_update_matches_reference_golden(name):
    """One bare nmf_algorithms_1D.update() from identical state (no clamp, no normalisation)."""
    from pydnmfk_amd.dist_nmf import nmf_algorithms_1D
    meta, A, W0, H0, z = load_case(name)
    if meta.get("prune"):
        pytest.skip("pruned cases are pinned through fit()")
    tol_step = _tols(meta)[0]
    args = _args(meta["k"], 1, meta["norm"], meta["W_update"], meta.get("method", "mu"))
    args.m, args.n, args.eps = meta["m"], meta["n"], float(np.finfo(np.float32).eps)
    W, H = torch.from_numpy(W0).cuda(), torch.from_numpy(H0).cuda()
    W1, H1 = nmf_algorithms_1D(torch.from_numpy(A).cuda(), W, H, params=args).update()
    assert W1 is W and H1 is H                                           # in place, same objects returned
    assert rel_fro(W.cpu().numpy(), z["r0_step1_W"]) <= tol_step
    assert rel_fro(H.cpu().numpy(), z["r0_step1_H"]) <= tol_step


@pytest.mark.parametrize("m,n,k,norm,itr", [(4096, 1024, 64, "fro", 20), (3000, 1500, 32, "fro", 20),
                                            (2048, 768, 128, "fro", 12), (2048, 1024, 64, "kl", 12),
                                            (1100, 900, 16, "kl", 12)])
def test_fit_matches_oracle_seeded(m, n, k, norm, itr):
    """Low-rank-plus-noise X (SURVEY 8d parity generator, RandomState(100) as in tests/test_dist_nmf_1d.py:14)."""
    from oracle import nmf_oracle as orc
    from pydnmfk_amd.pyDNMF import PyNMF
    rs = np.random.RandomState(100)
    A = np.abs(rs.rand(m, k) @ rs.rand(k, n) + 0.01 * rs.randn(m, n)).astype(np.float32)
    W0, H0 = rs.rand(m, k).astype(np.float32), rs.rand(k, n).astype(np.float32)
    W, H, err = PyNMF(torch.from_numpy(A).cuda(), factors=[W0, H0], params=_args(k, itr, norm)).fit()
    assert isinstance(W, torch.Tensor) and W.is_cuda                     # torch in -> torch out
    Wr, Hr, err_r = orc.fit_single(A, W0, H0, itr, norm=norm)
    assert rel_fro(W.cpu().numpy(), Wr) <= 1e-4
    assert rel_fro(H.cpu().numpy(), Hr) <= 1e-4
    assert abs(err - err_r) <= 1e-5


def test_reference_convergence_threshold():
    """The reference's own assertion (tests/test_dist_nmf_1d.py:46): rel_error < 1e-3 on the exact rank-2
    24x12 problem after 2000 MU iterations from a random init (here in float32 on the GPU)."""
    from pydnmfk_amd.pyDNMF import PyNMF
    np.random.seed(100)
    m, k, n = 24, 2, 12
    A = (np.random.rand(m, k) @ np.random.rand(k, n)).astype(np.float32)
    for norm, method in (("fro", "mu"), ("kl", "mu"), ("fro", "hals")):
        np.random.seed(5)
        _, _, err = PyNMF(A, factors=None, params=_args(k, 2000, norm, method=method)).fit()
        assert err < 1e-3, (norm, method)


def test_config2_size_properties():
    """BASELINE config 2 (65536 x 4096, k=32, MU/FRO, 1 GPU) -- too big for the CPU oracle in seconds, so
    checked through properties: non-negativity, monotone decrease of the Frobenius objective (the MU
    guarantee), unit column sums of W after normalisation, and a 'checksum of checksums' tying the
    contraction kernels to the residual kernel:
        ||A - WH||^2 = ||A||^2 - 2 <W^T A, H> + <W^T W, H H^T>."""
    from pydnmfk_amd.engine import HIP_OPS as ops, new_gram
    from pydnmfk_amd.pyDNMF import PyNMF
    m, n, k = 65536, 4096, 32
    g = torch.Generator(device="cuda").manual_seed(1234)
    A = torch.rand(m, n, device="cuda", generator=g)
    W0 = torch.rand(m, k, device="cuda", generator=g)
    H0 = torch.rand(k, n, device="cuda", generator=g)
    errs = []
    for itr in (1, 3, 6):
        W, H, err = PyNMF(A, factors=[W0, H0], params=_args(k, itr, "fro")).fit()
        errs.append(err)
        assert float(W.min()) >= 0 and float(H.min()) >= 0
        assert torch.isfinite(W).all() and torch.isfinite(H).all()
    assert errs[0] > errs[1] > errs[2]
    assert abs(float(W.sum(0).max()) - 1) < 1e-4 and abs(float(W.sum(0).min()) - 1) < 1e-4
    dev = A.device
    AtW = ops.wta(A, W, torch.empty(k, n, device=dev)).double()
    WtW = ops.gram_wtw(W, new_gram(k, dev))[:k, :k].double()
    HHt = ops.gram_hht(H, new_gram(k, dev))[:k, :k].double()
    a2 = float(ops.sqnorm(A))
    direct = float(ops.resid_sqnorm(A, W, H))
    identity = a2 - 2 * float((AtW * H.double()).sum()) + float((WtW * HHt).sum())
    assert abs(direct - identity) / a2 < 1e-6
    # A H^T against W^T A through <A H^T, W> = <W^T A, H>
    AH = ops.aht(A, H, torch.empty(m, k, device=dev)).double()
    lhs, rhs = float((AH * W.double()).sum()), float((AtW * H.double()).sum())
    assert abs(lhs - rhs) / abs(rhs) < 1e-6


def test_config3_size_row_sharding_and_checksums():
    """BASELINE config 3 at FULL size (262144 x 8192 fp32 = 8.6 GB, k = 64: offsets beyond 4 GiB) on one GPU.
    (1) What 8 ranks of a p_r = 8 grid compute -- every rank its 32768-row shard, the exchanged quantity being the sum of
    the per-shard [W^T A | W^T W] -- is replayed shard by shard through the same entry points and must equal the
    single-shot step on the whole matrix (the decomposition the RCCL allreduce implements; only fp32 summation order
    differs).  (2) The checksum-of-checksums identity and the adjoint identity of test_config2_size_properties hold at
    this size, i.e. no kernel mis-addresses a matrix larger than 2^32 bytes."""
    from pydnmfk_amd.engine import HIP_OPS as ops, new_gram
    m, n, k, P = 262144, 8192, 64, 8
    EPS = float(np.finfo(np.float32).eps)
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(7)
    A = torch.rand(m, n, device=dev, generator=g)
    W0 = torch.rand(m, k, device=dev, generator=g)
    H0 = torch.rand(k, n, device=dev, generator=g)
    # single shot
    W1, H1 = W0.clone(), H0.clone()
    ops.mu_fro_step(A, W1, H1, EPS, True, False)
    # the same step as 8 row shards + a sum
    W2, H2 = W0.clone(), H0.clone()
    G = ops.gram_hht(H2, new_gram(k, dev))
    ms = m // P
    AtW, WtW = torch.zeros(k, n, device=dev), torch.zeros_like(G)
    for s in range(P):
        ops.aht_update_w(A[s * ms:(s + 1) * ms], H2, G, W2[s * ms:(s + 1) * ms], EPS)
    for s in range(P):
        AtW += ops.wta(A[s * ms:(s + 1) * ms], W2[s * ms:(s + 1) * ms], torch.empty(k, n, device=dev))
        WtW += ops.gram_wtw(W2[s * ms:(s + 1) * ms], new_gram(k, dev))
    ops.mu_update_h(H2, AtW, WtW, EPS, False)
    assert float((W1 - W2).norm() / W1.norm()) < 1e-6
    assert float((H1 - H2).norm() / H1.norm()) < 1e-5
    assert torch.isfinite(W1).all() and torch.isfinite(H1).all() and float(W1.min()) >= 0 and float(H1.min()) >= 0
    # checksums at this size (W scaled so that the residual is not dominated by W H)
    W, H = W1 * 0.02, H1
    AtWd = ops.wta(A, W, torch.empty(k, n, device=dev)).double()
    WtWd = ops.gram_wtw(W, new_gram(k, dev))[:k, :k].double()
    HHt = ops.gram_hht(H, new_gram(k, dev))[:k, :k].double()
    a2 = float(ops.sqnorm(A))
    assert abs(a2 / (m * n / 3.0) - 1) < 1e-3                      # E[u^2] = 1/3 for U[0,1)
    direct = float(ops.resid_sqnorm(A, W, H))
    identity = a2 - 2 * float((AtWd * H.double()).sum()) + float((WtWd * HHt).sum())
    assert abs(direct - identity) / a2 < 1e-6
    AH = ops.aht(A, H, torch.empty(m, k, device=dev)).double()
    lhs, rhs = float((AH * W.double()).sum()), float((AtWd * H.double()).sum())
    assert abs(lhs - rhs) / abs(rhs) < 1e-6
    # the last rows / columns are really reached: a spike in the far corner must show up in both contractions
    A[m - 1, n - 1] += 1000.0
    AH2 = ops.aht(A, H, torch.empty(m, k, device=dev))
    assert torch.allclose((AH2[m - 1] - AH[m - 1].float()), 1000.0 * H[:, n - 1], rtol=1e-3, atol=1e-2)
    AtW2 = ops.wta(A, W, torch.empty(k, n, device=dev))
    assert torch.allclose((AtW2[:, n - 1] - AtWd[:, n - 1].float()), 1000.0 * W[m - 1], rtol=1e-3, atol=1e-2)


def test_config4_block_kl_checksums():
    """One rank's block of BASELINE config 4 (131072 x 65536 on a 4 x 2 grid -> 32768 x 32768 fp32 = 4 GiB, k = 128, KL).
    With U = A / (W H + eps) both KL products contract the same U against W H:
        <U H^T, W> = <W^T U, H> = sum(U * (W H)) = sum(A * WH / (WH + eps)) ~= sum(A),
    which ties dnmf_kl_uht and dnmf_kl_wtu to each other and to a plain sum at a size whose offsets exceed 2^32 bytes;
    then one whole MU/KL step must keep the factors finite, non-negative and lower the generalised KL divergence."""
    from pydnmfk_amd.engine import HIP_OPS as ops
    m, n, k = 32768, 32768, 128
    EPS = float(np.finfo(np.float32).eps)
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(11)
    A = torch.rand(m, n, device=dev, generator=g)
    W = torch.rand(m, k, device=dev, generator=g)
    H = torch.rand(k, n, device=dev, generator=g)
    UHT = ops.kl_uht(A, W, H, EPS, torch.empty(m, k, device=dev))
    WTU = ops.kl_wtu(A, W, H, EPS, torch.empty(k, n, device=dev))
    lhs, rhs = float((UHT.double() * W.double()).sum()), float((WTU.double() * H.double()).sum())
    sa = float(A.double().sum())
    assert abs(lhs - rhs) / abs(rhs) < 1e-6
    assert abs(lhs - sa) / sa < 1e-5

    def kl_div(W, H):          # sum(A log(A / WH) - A + WH), chunked over rows (the product is 4 GiB)
        tot = 0.0
        for r0 in range(0, m, 4096):
            a = A[r0:r0 + 4096]
            wh = W[r0:r0 + 4096] @ H
            tot += float((torch.xlogy(a, a / (wh + EPS)) - a + wh).double().sum())
        return tot

    d0 = kl_div(W, H)
    ops.mu_kl_step(A, W, H, EPS, True, False)
    assert torch.isfinite(W).all() and torch.isfinite(H).all() and float(W.min()) >= 0 and float(H.min()) >= 0
    d1 = kl_div(W, H)
    assert d1 < d0


def test_config4_whole_matrix_kl_checksums_on_one_gpu():
    """ALL of BASELINE config 4 on one GPU (131072 x 65536 fp32 = 34 GB, 8.6e9 elements: element indices beyond 2^32, row
    chunks beyond the 2 GiB descriptor window -- round 4: the chunk of `W^T U` had left the window and silently taken the slow
    path) -- what `bench.py --config 4` steps at N = 1.  The same identities as on the block: <U H^T, W> = <W^T U, H> ~= sum(A),
    plus a far-corner spike that must reach both products."""
    from pydnmfk_amd.engine import HIP_OPS as ops
    m, n, k = 131072, 65536, 128
    if torch.cuda.get_device_properties(0).total_memory < 60 * 2**30:
        pytest.skip("needs 36 GB of device memory")
    EPS = float(np.finfo(np.float32).eps)
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(12)
    A = torch.rand(m, n, device=dev, generator=g)
    W = torch.rand(m, k, device=dev, generator=g)
    H = torch.rand(k, n, device=dev, generator=g)
    UHT = ops.kl_uht(A, W, H, EPS, torch.empty(m, k, device=dev))
    WTU = ops.kl_wtu(A, W, H, EPS, torch.empty(k, n, device=dev))
    lhs, rhs = float((UHT.double() * W.double()).sum()), float((WTU.double() * H.double()).sum())
    sa = sum(float(A[r0:r0 + 8192].double().sum()) for r0 in range(0, m, 8192))
    assert abs(lhs - rhs) / abs(rhs) < 1e-6
    assert abs(lhs - sa) / sa < 1e-5
    # the last element of the matrix (linear index 2^33 - 1): raise it and watch the last row of U H^T / last column of W^T U move
    wh = float(W[m - 1].double() @ H[:, n - 1].double()) + EPS
    A[m - 1, n - 1] += 1000.0
    UHT2 = ops.kl_uht(A, W, H, EPS, torch.empty(m, k, device=dev))
    WTU2 = ops.kl_wtu(A, W, H, EPS, torch.empty(k, n, device=dev))
    assert torch.allclose(UHT2[m - 1] - UHT[m - 1], (1000.0 / wh) * H[:, n - 1], rtol=2e-3, atol=1e-2)
    assert torch.allclose(WTU2[:, n - 1] - WTU[:, n - 1], (1000.0 / wh) * W[m - 1], rtol=2e-3, atol=1e-2)
    assert torch.equal(UHT2[: m - 1], UHT[: m - 1]) and torch.equal(WTU2[:, : n - 1], WTU[:, : n - 1])


@pytest.mark.parametrize("k", [16, 9, 32])
def test_headline_size_kl_checksums_small_rank(k):
    """The KL products at the full 262144 x 8192 (8 GiB: offsets beyond 2^32 bytes, MUBUF windows re-based per chunk / per
    workgroup) for the ranks an NMFk sweep visits -- k = 16 (the 16-wide kernels on the factors as they are), k = 9 (through
    the zero-padded 16-wide images), k = 32 (the 32-wide kernels' MUBUF paths) -- tied to each other and to a plain sum:
        <U H^T, W> = <W^T U, H> = sum(A * WH / (WH + eps)) ~= sum(A);
    a spike in the far corner of A must reach both products (the last rows / columns are really read)."""
    from pydnmfk_amd.engine import HIP_OPS as ops
    m, n = 262144, 8192
    EPS = float(np.finfo(np.float32).eps)
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(13 + k)
    A = torch.rand(m, n, device=dev, generator=g)
    W = torch.rand(m, k, device=dev, generator=g)
    H = torch.rand(k, n, device=dev, generator=g)
    UHT = ops.kl_uht(A, W, H, EPS, torch.empty(m, k, device=dev))
    WTU = ops.kl_wtu(A, W, H, EPS, torch.empty(k, n, device=dev))
    lhs, rhs = float((UHT.double() * W.double()).sum()), float((WTU.double() * H.double()).sum())
    sa = float(A.double().sum())
    assert abs(lhs - rhs) / abs(rhs) < 1e-6
    assert abs(lhs - sa) / sa < 1e-5
    # far-corner spike: U[m-1][n-1] grows by spike / (W H + eps)[m-1][n-1]
    spike = 1000.0
    A[m - 1, n - 1] += spike
    UHT2 = ops.kl_uht(A, W, H, EPS, torch.empty(m, k, device=dev))
    WTU2 = ops.kl_wtu(A, W, H, EPS, torch.empty(k, n, device=dev))
    d = spike / (float(W[m - 1].double() @ H[:, n - 1].double()) + EPS)
    assert torch.allclose((UHT2[m - 1] - UHT[m - 1]).double(), d * H[:, n - 1].double(), rtol=2e-3, atol=1e-2)
    assert torch.allclose((WTU2[:, n - 1] - WTU[:, n - 1]).double(), d * W[m - 1].double(), rtol=2e-3, atol=1e-2)
    assert torch.equal(UHT2[: m - 1], UHT[: m - 1]) and torch.equal(WTU2[:, : n - 1], WTU[:, : n - 1])
