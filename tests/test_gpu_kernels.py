"""Per-kernel parity on the GPU: every C-ABI entry point against numpy (the same calls the reference
makes, cited per test) on seeded inputs, including ragged / unaligned shapes that force the generic
(non-vectorised) path and k that is not a multiple of the 32-wide MFMA tile.

Tolerance (fp32): rel-Frobenius <= 2e-6 per GEMM-like op vs a float64 numpy evaluation of the same
formula (fp32 MFMA is an exact fmaf chain; the difference is summation order), <= 1e-5 per fused update.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

EPS = float(np.finfo(np.float32).eps)
# (m, n, k): tile-aligned, ragged, tiny, k on both sides of the KP boundaries, unaligned leading dims
SHAPES = [(512, 384, 64), (300, 260, 32), (257, 131, 33), (130, 72, 128), (24, 12, 2), (1024, 256, 4),
          (97, 53, 7), (640, 200, 96), (2048, 1024, 64), (33, 515, 65),
          # rank <= 16 with whole k-tiles of columns: the 16x16x4 kernels (ragged rows, k = 1, k % 4 != 0, a single tile)
          (300, 256, 16), (1000, 512, 5), (129, 128, 1), (4100, 1024, 13), (257, 64, 16), (8192, 2048, 9)]


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
    A[rs.rand(m, n) < 0.2] = 0.0          # exact zeros, as in swim.mat
    W = rs.rand(m, k).astype(np.float32)
    H = rs.rand(k, n).astype(np.float32)
    return A, W, H


def _d(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


@pytest.mark.parametrize("m,n,k", SHAPES)
def test_grams(ops, m, n, k):
    """np.matmul(A.T, A) in global_gram (dist_nmf.py:679): H H^T and W^T W, zero-padded to KP x KP."""
    from pydnmfk_amd.engine import new_gram
    _, W, H = _mk(m, n, k)
    G = new_gram(k, torch.device("cuda"))
    G.fill_(7.0)
    g = ops.gram_hht(_d(H), G).cpu().numpy()
    assert _rel(g[:k, :k], H.astype(np.float64) @ H.T.astype(np.float64)) < 2e-6
    assert not g[k:].any() and not g[:, k:].any()
    assert np.array_equal(g, g.T)
    G.fill_(7.0)
    g = ops.gram_wtw(_d(W), G).cpu().numpy()
    assert _rel(g[:k, :k], W.T.astype(np.float64) @ W.astype(np.float64)) < 2e-6
    assert not g[k:].any() and not g[:, k:].any()


@pytest.mark.parametrize("m,n,k", SHAPES)
def test_aht_and_wta(ops, m, n, k):
    """np.matmul(A, H.T) and np.matmul(W.T, A) in global_mm (dist_nmf.py:705)."""
    A, W, H = _mk(m, n, k)
    dA = _d(A)
    AH = ops.aht(dA, _d(H), torch.full((m, k), 7.0, device="cuda")).cpu().numpy()
    assert _rel(AH, A.astype(np.float64) @ H.T.astype(np.float64)) < 2e-6
    AtW = ops.wta(dA, _d(W), torch.full((k, n), 7.0, device="cuda")).cpu().numpy()
    assert _rel(AtW, W.T.astype(np.float64) @ A.astype(np.float64)) < 2e-6


@pytest.mark.parametrize("m,n,k", [(89600, 96, 64), (163968, 40, 16), (82000, 36, 128)])
def test_tall_shards(ops, m, n, k):
    """Tall, narrow shards: many row tiles per launch, few k-tiles (the opposite corner from SHAPES)."""
    from pydnmfk_amd.engine import new_gram
    A, W, H = _mk(m, n, k)
    A64, W64, H64 = A.astype(np.float64), W.astype(np.float64), H.astype(np.float64)
    AH = ops.aht(_d(A), _d(H), torch.empty(m, k, device="cuda")).cpu().numpy()
    assert _rel(AH, A64 @ H64.T) < 2e-6
    G = ops.gram_hht(_d(H), new_gram(k, torch.device("cuda")))
    Wf = _d(W)
    ops.aht_update_w(_d(A), _d(H), G, Wf, EPS)
    assert _rel(Wf.cpu().numpy(), W64 * ((A64 @ H64.T) / (W64 @ (H64 @ H64.T) + EPS))) < 1e-5
    Wd = _d(W)
    ops.mu_update_w(Wd, _d(AH), G, EPS)
    assert _rel(Wd.cpu().numpy(), W64 * (AH.astype(np.float64) / (W64 @ (H64 @ H64.T) + EPS))) < 1e-5


def test_strided_views(ops):
    """Leading dimensions larger than the logical width (sub-blocks of bigger buffers)."""
    m, n, k = 200, 136, 64
    A, W, H = _mk(m, n, k)
    bigA = torch.zeros(m, n + 8, device="cuda"); bigA[:, :n] = _d(A)
    bigH = torch.zeros(k, n + 12, device="cuda"); bigH[:, :n] = _d(H)
    bigW = torch.zeros(m, k + 4, device="cuda"); bigW[:, :k] = _d(W)
    AH = ops.aht(bigA[:, :n], bigH[:, :n], torch.empty(m, k, device="cuda")).cpu().numpy()
    assert _rel(AH, A.astype(np.float64) @ H.T.astype(np.float64)) < 2e-6
    AtW = ops.wta(bigA[:, :n], bigW[:, :k], torch.empty(k, n, device="cuda")).cpu().numpy()
    assert _rel(AtW, W.T.astype(np.float64) @ A.astype(np.float64)) < 2e-6


@pytest.mark.parametrize("m,n,k", SHAPES)
def test_mu_updates(ops, m, n, k):
    """W *= AH / (W HHT + eps) (dist_nmf.py:731-732); H *= AtW / (H^T WTW + eps)^T (:750-751); fused W phase."""
    from pydnmfk_amd.engine import new_gram
    A, W, H = _mk(m, n, k)
    A64, W64, H64 = A.astype(np.float64), W.astype(np.float64), H.astype(np.float64)
    dev = torch.device("cuda")
    G = ops.gram_hht(_d(H), new_gram(k, dev))
    AH = (A64 @ H64.T).astype(np.float32)
    Wd = _d(W)
    ops.mu_update_w(Wd, _d(AH), G, EPS)
    W_ref = W64 * (AH.astype(np.float64) / (W64 @ (H64 @ H64.T) + EPS))
    assert _rel(Wd.cpu().numpy(), W_ref) < 1e-5
    Wf = _d(W)
    ops.aht_update_w(_d(A), _d(H), G, Wf, EPS)
    assert _rel(Wf.cpu().numpy(), W_ref) < 1e-5
    G2 = ops.gram_wtw(_d(W), new_gram(k, dev))
    AtW = (W64.T @ A64).astype(np.float32)
    for clamp in (False, True):
        Hd = _d(H)
        ops.mu_update_h(Hd, _d(AtW), G2, EPS, clamp)
        H_ref = H64 * (AtW.astype(np.float64) / (H64.T @ (W64.T @ W64) + EPS).T)
        if clamp:
            H_ref = np.maximum(H_ref, EPS)
        assert _rel(Hd.cpu().numpy(), H_ref) < 1e-5
        if clamp:
            assert float(Hd.min()) >= EPS


@pytest.mark.parametrize("m,n,k", SHAPES)
def test_kl_pieces(ops, m, n, k):
    """U = A/(WH+eps); U H^T; W^T U (dist_nmf.py:806-810); row/column sums (:793-795); eltwise (:828-830, :847-849)."""
    A, W, H = _mk(m, n, k)
    A64, W64, H64 = A.astype(np.float64), W.astype(np.float64), H.astype(np.float64)
    U = A64 / (W64 @ H64 + EPS)
    uht = ops.kl_uht(_d(A), _d(W), _d(H), EPS, torch.full((m, k), 7.0, device="cuda")).cpu().numpy()
    assert _rel(uht, U @ H64.T) < 5e-6
    wtu = ops.kl_wtu(_d(A), _d(W), _d(H), EPS, torch.full((k, n), 7.0, device="cuda")).cpu().numpy()
    assert _rel(wtu, W64.T @ U) < 5e-6
    x2 = ops.rowsum(_d(H), torch.empty(k, device="cuda")).cpu().numpy()
    assert _rel(x2, H64.sum(1)) < 1e-6
    x1 = ops.colsum(_d(W), torch.empty(k, device="cuda")).cpu().numpy()
    assert _rel(x1, W64.sum(0)) < 1e-6
    Wd = _d(W)
    ops.kl_update_w(Wd, _d(uht), _d(x2), EPS)
    assert _rel(Wd.cpu().numpy(), W64 * (uht / (x2[None, :].astype(np.float64) + EPS))) < 1e-6
    Hd = _d(H)
    ops.kl_update_h(Hd, _d(wtu), _d(x1), EPS, False)
    assert _rel(Hd.cpu().numpy(), H64 * (wtu / (x1[:, None].astype(np.float64) + EPS))) < 1e-6


@pytest.mark.parametrize("m,n,k,pad", [(500, 96, 12, 0), (163968, 64, 16, 0), (2000, 4096, 16, 0), (777, 640, 16, 8),
                                       (4096, 1024, 3, 4), (65, 32, 16, 0), (40000, 192, 14, 0)])
def test_kl16_products(ops, m, n, k, pad):
    """The 16-wide KL kernels (csrc/dnmf_kl16.h; dist_nmf.py:806-810): a column count that only the U H^T kernel takes
    (n % 32 == 0, n % 64 != 0), many row chunks / column splits, sub-blocks of wider buffers (`pad` extra columns in every
    leading dimension), ranks that go through the zero-padded factor images."""
    A, W, H = _mk(m, n, k)
    A64, W64, H64 = A.astype(np.float64), W.astype(np.float64), H.astype(np.float64)
    U = A64 / (W64 @ H64 + EPS)

    def view(x):
        r, c = x.shape
        big = torch.full((r, c + pad), 3.0, device="cuda")
        big[:, :c] = _d(x)
        return big[:, :c]

    dA, dW, dH = view(A), view(W), view(H)
    uht = torch.full((m, k + pad), 7.0, device="cuda")
    ops.kl_uht(dA, dW, dH, EPS, uht[:, :k])
    assert _rel(uht[:, :k].cpu().numpy(), U @ H64.T) < 5e-6
    wtu = torch.full((k, n + pad), 7.0, device="cuda")
    ops.kl_wtu(dA, dW, dH, EPS, wtu[:, :n])
    assert _rel(wtu[:, :n].cpu().numpy(), W64.T @ U) < 5e-6
    if pad:
        assert float(uht[:, k:].min()) == 7.0 and float(wtu[:, n:].min()) == 7.0      # nothing written beyond the logical width


@pytest.mark.parametrize("m,nb,nh,k", [(1000, 4, 96, 64), (300, 2, 32, 128), (2048, 4, 512, 16), (700, 3, 64, 5), (129, 2, 160, 33),
                                       (33000, 4, 2048, 128), (40000, 2, 1024, 16), (70000, 8, 1024, 64)])
def test_aht_with_h_as_column_blocks(ops, m, nb, nh, k):
    """dnmf_aht_hblocks: A H^T with H handed over as the stack of column blocks [nb][k][nh] an allgather of the ranks' slices
    leaves (AH_glob, dist_nmf.py:195-197) equals the product with the assembled H -- the two- / three-tile MUBUF loops (block
    offset = one scalar multiply-high per tile), the generic edge path, the 16-wide kernel, ranks below the padded width."""
    n = nb * nh
    A, W, H = _mk(m, n, k)
    Hs = _d(np.ascontiguousarray(H.reshape(k, nb, nh).transpose(1, 0, 2)))
    out = torch.full((m, k), 7.0, device="cuda")
    ops.aht_hblocks(_d(A), Hs, out)
    assert _rel(out.cpu().numpy(), A.astype(np.float64) @ H.T.astype(np.float64)) < 2e-6
    ref = ops.aht(_d(A), _d(H), torch.empty(m, k, device="cuda"))
    assert torch.equal(out, ref)                     # same kernel, same order of summation: only the addresses differ


@pytest.mark.parametrize("m,nb,nh,k", [(1000, 4, 96, 64), (300, 2, 32, 128), (2048, 4, 512, 16), (700, 3, 64, 5), (129, 2, 160, 33),
                                       (33000, 4, 2048, 128), (40000, 2, 1024, 16)])
def test_kl_uht_with_h_as_column_blocks(ops, m, nb, nh, k):
    """dnmf_kl_uht_hblocks: H handed over as the stack of column blocks [nb][k][nh] that an allgather of the ranks' slices
    leaves (gather_W_H, dist_nmf.py:283-287) equals the product with the assembled H -- 32-wide and 16-wide kernels, ranks
    that go through the zero-padded images, several column splits per block."""
    n = nb * nh
    A, W, H = _mk(m, n, k)
    A64, W64, H64 = A.astype(np.float64), W.astype(np.float64), H.astype(np.float64)
    U = A64 / (W64 @ H64 + EPS)
    Hs = _d(np.ascontiguousarray(H.reshape(k, nb, nh).transpose(1, 0, 2)))
    out = torch.full((m, k), 7.0, device="cuda")
    ops.kl_uht_hblocks(_d(A), _d(W), Hs, EPS, out)
    assert _rel(out.cpu().numpy(), U @ H64.T) < 5e-6
    ref = ops.kl_uht(_d(A), _d(W), _d(H), EPS, torch.empty(m, k, device="cuda"))
    assert _rel(out.cpu().numpy(), ref.cpu().numpy()) < 2e-6


@pytest.mark.parametrize("m,n,k", SHAPES)
def test_norms_and_fit_helpers(ops, m, n, k):
    """np.linalg.norm(A)**2 and np.linalg.norm(A - W@H)**2 (pyDNMF.py:207-217); clamp (:155-157); normalise (:185-194)."""
    A, W, H = _mk(m, n, k)
    W *= 0.1
    A64, W64, H64 = A.astype(np.float64), W.astype(np.float64), H.astype(np.float64)
    assert abs(float(ops.sqnorm(_d(A))) / float((A64 ** 2).sum()) - 1) < 1e-6
    assert abs(float(ops.resid_sqnorm(_d(A), _d(W), _d(H))) / float(((A64 - W64 @ H64) ** 2).sum()) - 1) < 5e-6
    X = _d(A - 0.5)
    ops.clamp_min(X, EPS)
    assert np.array_equal(X.cpu().numpy(), np.maximum(A - 0.5, np.float32(EPS)))
    s = ops.colsum(_d(W), torch.empty(k, device="cuda"))
    Wd, Hd = _d(W), _d(H)
    ops.scale_cols_div(Wd, s, EPS)
    ops.scale_rows_mul(Hd, s)
    sn = s.cpu().numpy()
    assert _rel(Wd.cpu().numpy(), W / (sn[None, :] + np.float32(EPS))) < 1e-6
    assert _rel(Hd.cpu().numpy(), H * sn[:, None]) < 1e-6


@pytest.mark.parametrize("m,n,k", SHAPES)
@pytest.mark.parametrize("norm", ["fro", "kl"])
def test_whole_step_matches_oracle(ops, m, n, k, norm):
    """dnmf_mu_{fro,kl}_step vs the oracle's single-rank step (dist_nmf.py:755-771 / :851-869), with and
    without the W update (regression mode) and with the fit loop's clamp."""
    from oracle import nmf_oracle as orc
    A, W, H = _mk(m, n, k)
    for w_update, clamp in ((True, False), (False, False), (True, True)):
        Wd, Hd = _d(W), _d(H)
        (ops.mu_fro_step if norm == "fro" else ops.mu_kl_step)(_d(A), Wd, Hd, EPS, w_update, clamp)
        Wr, Hr = W.copy(), H.copy()
        step = orc.fro_mu_step_local if norm == "fro" else orc.kl_mu_step_local
        step(A, Wr, Hr, np.float32(EPS), W_update=w_update)
        if clamp:
            Wr, Hr = np.maximum(Wr, np.float32(EPS)), np.maximum(Hr, np.float32(EPS))
        assert _rel(Wd.cpu().numpy(), Wr) < 1e-5, (w_update, clamp)
        assert _rel(Hd.cpu().numpy(), Hr) < 1e-5, (w_update, clamp)


@pytest.mark.parametrize("m,n,k", SHAPES)
def test_hals_sweeps(ops, m, n, k):
    """HALS column sweeps (dist_nmf.py:884-891, :905-909) against the oracle's single-rank step, from identical state."""
    from oracle import nmf_oracle as orc
    from pydnmfk_amd.engine import new_gram
    A, W, H = _mk(m, n, k)
    dev = torch.device("cuda")
    # W sweep alone
    A64, H64 = A.astype(np.float64), H.astype(np.float64)
    AH = (A64 @ H64.T).astype(np.float32)
    G = ops.gram_hht(_d(H), new_gram(k, dev))
    Wd = _d(W)
    ops.hals_update_w(Wd, _d(AH), G, EPS)
    Wr = W.copy()
    HHT = np.matmul(H, H.T)
    for kk in range(k):
        t = Wr[:, kk] * HHT[kk, kk] + AH[:, kk] - Wr.dot(HHT[:, kk])
        Wr[:, kk] = np.maximum(t, np.float32(EPS))
        ss = np.linalg.norm(Wr[:, kk])
        if ss > 0:
            Wr[:, kk] /= ss
    assert _rel(Wd.cpu().numpy(), Wr) < 2e-4          # cancellation-prone numerator, see tests/test_oracle_golden.py
    assert np.allclose(np.linalg.norm(Wd.cpu().numpy().astype(np.float64), axis=0), 1.0, atol=1e-5)
    # column-by-column form (the multi-rank path) gives the same result as the one-call form
    Wc = _d(W)
    ss2 = ops.hals_ss2(k, Wc)
    for kk in range(k):
        ops.hals_w_col(Wc, _d(AH), G, kk, ss2, EPS)
    ops.hals_w_scale(Wc, k - 1, ss2)
    # (the one-call form keeps the rows in registers and drops the cancelling pair W[i][kk] G[kk][kk] - W[i][kk] G[kk][kk]
    #  of dist_nmf.py:887; the column kernels evaluate the reference's expression as written)
    assert _rel(Wc.cpu().numpy(), Wd.cpu().numpy()) < 2e-4
    assert _rel(Wc.cpu().numpy(), Wr) < 2e-4
    # and the k-launch composite the persistent sweep falls back to equals the explicit column calls bit for bit
    We = _d(W)
    ops.hals_update_w_columns(We, _d(AH), G, EPS)
    assert torch.equal(We, Wc)
    # H sweep alone
    W64 = Wr.astype(np.float64)
    AtW = (W64.T @ A64).astype(np.float32)
    G2 = ops.gram_wtw(_d(Wr), new_gram(k, dev))
    Hd = _d(H)
    ops.hals_update_h(Hd, _d(AtW), G2, EPS)
    Hr = H.copy()
    WTW = np.matmul(Wr.T, Wr)
    for kk in range(k):
        t = Hr[kk, :] + AtW[kk, :] - WTW[kk, :].dot(Hr)
        Hr[kk, :] = np.maximum(t, np.float32(EPS))
    assert _rel(Hd.cpu().numpy(), Hr) < 2e-4
    assert float(Hd.min()) >= EPS


@pytest.mark.parametrize("m,k", [(70000, 64), (262144, 16), (131072, 64), (40001, 5), (9000, 128), (300000, 64)])
def test_hals_persistent_w_sweep_large(ops, m, k):
    """The one-launch W sweep (one lane per row, grid-wide column norms through per-workgroup slots) on grids of many
    workgroups -- 300000 x 64 exceeds what the device keeps resident and takes the column path -- against a float64
    evaluation of dist_nmf.py:884-891, and bit-reproducible from run to run."""
    from pydnmfk_amd.engine import new_gram
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(k + m)
    W = torch.rand(m, k, device=dev, generator=g)
    H = torch.rand(k, 512, device=dev, generator=g)
    AH = torch.rand(m, k, device=dev, generator=g) * 200.0
    G = ops.gram_hht(H, new_gram(k, dev))
    outs = []
    for _ in range(2):
        Wd = W.clone()
        ops.hals_update_w(Wd, AH, G, EPS)
        outs.append(Wd)
    assert torch.equal(outs[0], outs[1])
    Wr = W.double()
    G64, A64 = G[:k, :k].double(), AH.double()
    for kk in range(k):
        t = Wr[:, kk] * G64[kk, kk] + A64[:, kk] - Wr @ G64[:, kk]
        Wr[:, kk] = torch.clamp(t, min=EPS)
        ss = torch.linalg.norm(Wr[:, kk])
        if float(ss) > 0:
            Wr[:, kk] /= ss
    assert float((outs[0].double() - Wr).norm() / Wr.norm()) < 5e-5
    assert torch.allclose(outs[0].double().norm(dim=0), torch.ones(k, dtype=torch.float64, device=dev), atol=1e-5)


def test_hals_sweep_status_tracks_lost_coresidency(ops):
    """Two persistent W sweeps enqueued on two streams, each large enough to want the whole device: whichever way the
    dispatcher interleaves them, the sticky status word must tell the truth -- set exactly when a sweep gave up waiting
    (its W is then NaN), clear when both finished normally (their W then equals the per-column sweep) -- and
    HipOps.hals_check must raise on it.  The query clears the word."""
    from pydnmfk_amd._lib import DnmfError
    from pydnmfk_amd.engine import new_gram
    import ctypes
    from pydnmfk_amd._lib import lib
    dev = torch.device("cuda")
    m, k = 262144, 64
    g = torch.Generator(device="cuda").manual_seed(5)
    H = torch.rand(k, 512, device=dev, generator=g)
    G = ops.gram_hht(H, new_gram(k, dev))
    Ws = [torch.rand(m, k, device=dev, generator=g) for _ in range(2)]
    AHs = [torch.rand(m, k, device=dev, generator=g) * 200.0 for _ in range(2)]
    refs = []
    for W, AH in zip(Ws, AHs):
        Wc = W.clone()
        ops.hals_update_w_columns(Wc, AH, G, EPS)
        refs.append(Wc)
    ops.hals_check()                                                   # nothing has timed out so far
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    for st, W, AH in zip(streams, Ws, AHs):
        with torch.cuda.stream(st):
            ops.hals_update_w(W, AH, G, EPS)
    torch.cuda.synchronize()
    lost = bool(torch.isnan(Ws[0]).any() or torch.isnan(Ws[1]).any())
    if lost:
        with pytest.raises(DnmfError, match="co-residency|resident"):
            ops.hals_check()
    else:
        ops.hals_check()
        for W, R in zip(Ws, refs):
            assert float((W - R).norm() / R.norm()) < 2e-4
    flag = ctypes.c_int(7)
    assert lib.dnmf_hals_sweep_status(ctypes.byref(flag), None) == 0 and flag.value == 0      # cleared by the query above


def test_errors_are_loud(ops):
    from pydnmfk_amd._lib import DnmfError
    with pytest.raises(TypeError):
        ops.aht(torch.rand(4, 4), torch.rand(2, 4), torch.empty(4, 2))           # CPU tensors
    with pytest.raises(TypeError):
        ops.sqnorm(torch.rand(4, 4, device="cuda", dtype=torch.float64))        # float64
    with pytest.raises((ValueError, DnmfError)):
        ops.gram_wtw(torch.rand(300, 300, device="cuda"), torch.empty(256, 256, device="cuda"))  # k > 256
    with pytest.raises(ValueError):     # a Gram buffer smaller than the KP x KP block the library writes (k = 200 -> KP = 256)
        ops.gram_wtw(torch.rand(300, 200, device="cuda"), torch.empty(128, 128, device="cuda"))
    with pytest.raises(ValueError):
        ops.gram_hht(torch.rand(40, 300, device="cuda"), torch.empty(32, 32, device="cuda"))


def test_clock_probe_reads_a_plausible_clock():
    """dnmf_clock_probe (measurement aid): s_memtime against the 100 MHz wall clock while a kernel runs -- a gfx950 shader clock
    lies between 0.1 and 2.6 GHz, and the samples are ordered in time."""
    from pydnmfk_amd.engine import ClockProbe, HIP_OPS as ops
    A = torch.rand(8192, 4096, device="cuda")
    pr = ClockProbe(5.0)
    for _ in range(20):
        ops.sqnorm(A)
    torch.cuda.synchronize()
    t, ghz = pr.samples()
    assert t.numel() >= 8 and bool((t[1:] > t[:-1]).all())
    g = pr.held_ghz()
    assert g is not None and 0.1 < g < 2.6, g
    with pytest.raises(Exception):
        from pydnmfk_amd._lib import lib, check
        check(lib.dnmf_clock_probe(0, 4, 1, 0))


@pytest.mark.parametrize("m,n,k", [(1, 64, 1), (37, 64, 3), (300, 128, 8), (1000, 256, 16), (4100, 512, 16), (70000, 4096, 16),
                                   (65, 100, 5), (513, 130, 16), (257, 64, 17), (1000, 260, 24), (5000, 512, 32), (66000, 4096, 32),
                                   (300, 256, 33), (2000, 512, 64), (300, 128, 128)])
@pytest.mark.parametrize("bf16", [False, True])
def test_wta_gram_is_wta_plus_gram(m, n, k, bf16):
    """dnmf_wta_gram: W^T A bit-identical to dnmf_wta (the same kernels compute it), W^T W equal to the float64 product and
    zero padded -- the riding Gram accumulator of the k <= 32 kernels (16-wide, 32-wide, vector and generic paths, ragged row
    chunks), and the two-call form above k = 32 (dist_nmf.py:705, :747-748)."""
    from pydnmfk_amd.engine import HIP_OPS as ops, new_gram
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(m * 7 + n + k)
    A = torch.rand(m, n, device=dev, generator=g)
    if bf16:
        A = A.to(torch.bfloat16)
    W = torch.rand(m, k, device=dev, generator=g)
    ref = ops.wta(A, W, torch.empty(k, n, device=dev))
    G = torch.full_like(new_gram(k, dev), 7.0)
    out = ops.wta_gram(A, W, torch.full((k, n), 3.0, device=dev), G)
    assert torch.equal(out, ref)
    G64 = W.double().t() @ W.double()
    assert float((G[:k, :k].double() - G64).norm() / G64.norm()) < 2e-6
    Z = G.clone()
    Z[:k, :k] = 0
    assert float(Z.abs().max()) == 0.0
    assert torch.equal(G[:k, :k], G[:k, :k].t())        # the same products in the same order on both sides of the diagonal
    G2 = torch.full_like(G, 5.0)
    ops.wta_gram(A, W, torch.empty(k, n, device=dev), G2)
    assert torch.equal(G, G2)                             # run-to-run bit reproducible


def test_perturb_uniform_is_the_reference_distribution(ops):
    """dnmf_perturb_uniform = NMFk's `sample.randM` (pyDNMFk.py:42-44) in one pass: X * (1 + nv + 2 nv U), U ~ U[0,1) from a counter-based
    generator.  The ratio X_per / X must be uniform on [1 + nv, 1 + 3 nv) (range, mean, variance, no correlation between
    neighbours), a function of (seed, position) only, and the bf16 form must equal the fp32 form rounded once."""
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(3)
    m, n, nv = 2048, 1024, 0.03
    X = torch.rand(m, n, device=dev, generator=g) + 0.5
    Y = ops.perturb_uniform(X, nv, 1000)
    assert Y is not None and Y.dtype == torch.float32 and Y.shape == X.shape
    ratio = (Y.double() / X.double())
    u = (ratio - 1 - nv) / (2 * nv)                                   # should be U[0, 1)
    assert float(u.min()) > -1e-5 and float(u.max()) < 1 + 1e-5
    N = u.numel()
    assert abs(float(u.mean()) - 0.5) < 5 * (1 / 12) ** 0.5 / N ** 0.5
    assert abs(float(u.var()) - 1 / 12) < 5e-4
    uc = u - u.mean()
    for a, b in ((uc[:, :-1], uc[:, 1:]), (uc[:-1], uc[1:])):        # neighbours along a row / down a column
        assert abs(float((a * b).mean()) / float(uc.var())) < 5e-3
    hist = torch.histc(u.float(), bins=64, min=0.0, max=1.0)
    assert float((hist - N / 64).abs().max()) < 6 * (N / 64) ** 0.5
    assert torch.equal(Y, ops.perturb_uniform(X, nv, 1000))           # a function of (seed, position)
    Y2 = ops.perturb_uniform(X, nv, 2000)
    assert float((Y2 != Y).float().mean()) > 0.99
    # a view with a row pitch reads the same values as its contiguous copy (the uniforms are keyed by position in the matrix)
    big = torch.zeros(m, n + 64, device=dev)
    big[:, :n] = X
    assert torch.equal(ops.perturb_uniform(big[:, :n], nv, 1000), Y)
    # bf16 storage: the same uniforms, scaled in fp32, rounded once to nearest even
    Xb = X.to(torch.bfloat16)
    Yb = ops.perturb_uniform(Xb, nv, 1000)
    assert Yb.dtype == torch.bfloat16
    assert torch.equal(Yb, ops.perturb_uniform(Xb.float(), nv, 1000).to(torch.bfloat16))
    # ANY shape: a block whose rows are not whole aligned 8-element vectors takes the element-per-thread kernel, which gives every
    # element the value the vector kernel gives it -- one random stream per seed whatever the block's shape.  A view
    # with an odd pitch / an unaligned base of the SAME logical matrix must therefore reproduce Y exactly
    odd = torch.zeros(m, n + 3, device=dev)
    odd[:, 1:n + 1] = X
    assert torch.equal(ops.perturb_uniform(odd[:, 1:n + 1], nv, 1000), Y)
    Z = ops.perturb_uniform(X[:, :1001].contiguous(), nv, 1)         # cols % 8 != 0
    rz = (Z.double() / X[:, :1001].double() - 1 - nv) / (2 * nv)
    assert Z.shape == (m, 1001) and float(rz.min()) > -1e-5 and float(rz.max()) < 1 + 1e-5 and abs(float(rz.mean()) - 0.5) < 2e-3
    oddb = odd.to(torch.bfloat16)
    assert torch.equal(ops.perturb_uniform(oddb[:, 1:n + 1], nv, 1000), Yb)
    # strided columns are left to the caller
    assert ops.perturb_uniform(X[:, ::2], nv, 1) is None
