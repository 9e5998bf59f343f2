"""Seeded random shapes through the big entry points (every dispatch branch: 16-wide / 32-wide kernels, vector /
generic paths, interior / edge tiles, fp32 / bf16-stored X) against a float64 numpy evaluation of the reference
formulas (dist_nmf.py:705, 716-751).  Tolerances as in test_gpu_kernels.py."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

EPS = float(np.finfo(np.float32).eps)


def _rel(x, ref):
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.linalg.norm(np.asarray(x, dtype=np.float64) - ref) / max(np.linalg.norm(ref), 1e-300))


def _shapes(seed, count):
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(count):
        k = int(rs.choice([1, 2, 3, 4, 5, 8, 11, 16, 17, 24, 32, 33, 48, 64, 65, 96, 128]))
        m = int(rs.choice([1, 7, 31, 32, 33, 100, 127, 128, 129, 255, 256, 257, 500, 1000, 1025, 2049]))
        n = int(rs.choice([4, 12, 31, 32, 33, 60, 64, 96, 100, 128, 129, 192, 256, 260, 384, 512, 1000, 1024]))
        out.append((m, n, k, bool(rs.rand() < 0.4), int(rs.randint(1 << 30))))
    return out


import os  # noqa: E402  (DNMF_FUZZ="seed,count" widens the sweep for ad-hoc runs)

_SEED, _COUNT = (int(x) for x in os.environ.get("DNMF_FUZZ", "2024,160").split(","))


@pytest.mark.parametrize("m,n,k,bf16,seed", _shapes(_SEED, _COUNT))
def test_random_shapes(m, n, k, bf16, seed):
    from pydnmfk_amd.engine import HIP_OPS as ops, new_gram
    rs = np.random.RandomState(seed)
    A = rs.rand(m, n).astype(np.float32)
    A[rs.rand(m, n) < 0.1] = 0.0
    W = rs.rand(m, k).astype(np.float32)
    H = rs.rand(k, n).astype(np.float32)
    dev = torch.device("cuda")
    dA = torch.from_numpy(A).to(dev)
    if bf16:
        dA = dA.to(torch.bfloat16)
        A = dA.float().cpu().numpy()
    A64, W64, H64 = A.astype(np.float64), W.astype(np.float64), H.astype(np.float64)
    dW, dH = torch.from_numpy(W).to(dev), torch.from_numpy(H).to(dev)
    AH = ops.aht(dA, dH, torch.full((m, k), 3.0, device=dev)).cpu().numpy()
    assert _rel(AH, A64 @ H64.T) < 2e-6
    AtW = ops.wta(dA, dW, torch.full((k, n), 3.0, device=dev)).cpu().numpy()
    assert _rel(AtW, W64.T @ A64) < 2e-6
    G = ops.gram_hht(dH, new_gram(k, dev))
    Wf = dW.clone()
    ops.aht_update_w(dA, dH, G, Wf, EPS)
    W_ref = W64 * ((A64 @ H64.T) / (W64 @ (H64 @ H64.T) + EPS))
    assert _rel(Wf.cpu().numpy(), W_ref) < 1e-5
    Ws, Hs = dW.clone(), dH.clone()
    ops.mu_fro_step(dA, Ws, Hs, EPS, True, False)
    H_ref = H64 * ((W_ref.T @ A64) / ((W_ref.T @ W_ref) @ H64 + EPS))
    assert _rel(Ws.cpu().numpy(), W_ref) < 1e-5
    assert _rel(Hs.cpu().numpy(), H_ref) < 2e-5
    a2 = max(float((A64 ** 2).sum()), 1e-30)       # the residual itself can be ~0 (m = 1): compare on the scale of ||A||^2
    assert abs(float(ops.resid_sqnorm(dA, Ws, Hs)) - float(((A64 - W_ref @ H_ref) ** 2).sum())) / a2 < 2e-5
    assert abs(float(ops.sqnorm(dA)) / a2 - 1) < 1e-6


def _shapes_kl(seed, count):
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(count):
        k = int(rs.choice([1, 3, 4, 8, 16, 20, 32, 40, 64, 72, 128]))
        m = int(rs.choice([5, 32, 33, 96, 128, 130, 256, 300, 640, 1024, 1500]))
        n = int(rs.choice([8, 31, 32, 64, 100, 128, 160, 256, 300, 512, 1024, 1100]))
        out.append((m, n, k, int(rs.randint(1 << 30))))
    return out


@pytest.mark.parametrize("m,n,k,seed", _shapes_kl(_SEED + 1, max(40, _COUNT // 2)))
def test_random_shapes_kl_and_hals(m, n, k, seed):
    """Whole MU/KL and HALS/FRO steps (dist_nmf.py:851-869, 873-934) against the oracle's single-rank steps."""
    from oracle import nmf_oracle as orc
    from pydnmfk_amd.dist_comm import MPI_comm
    from pydnmfk_amd.dist_nmf import nmf_algorithms_1D
    from pydnmfk_amd.engine import HIP_OPS as ops
    from pydnmfk_amd.utils import parse
    rs = np.random.RandomState(seed)
    A = (rs.rand(m, n) + 0.05).astype(np.float32)
    A[rs.rand(m, n) < 0.1] = 0.0
    W = (rs.rand(m, k) + 0.05).astype(np.float32)
    H = (rs.rand(k, n) + 0.05).astype(np.float32)
    dev = torch.device("cuda")
    dA = torch.from_numpy(A).to(dev)
    Wd, Hd = torch.from_numpy(W).to(dev), torch.from_numpy(H).to(dev)
    ops.mu_kl_step(dA, Wd, Hd, EPS, True, False)
    Wr, Hr = W.copy(), H.copy()
    orc.kl_mu_step_local(A, Wr, Hr, np.float32(EPS), W_update=True)
    assert _rel(Wd.cpu().numpy(), Wr) < 1e-5 and _rel(Hd.cpu().numpy(), Hr) < 1e-5
    comms = MPI_comm(None, 1, 1)
    p = parse()
    p.comm1, p.comm, p.p_r, p.p_c, p.k, p.m, p.n = comms.comm, comms, 1, 1, k, m, n
    p.row_comm, p.col_comm = comms.cart_1d_row(), comms.cart_1d_column()
    p.norm, p.method, p.W_update, p.eps = "fro", "hals", True, EPS
    Wd, Hd = torch.from_numpy(W).to(dev), torch.from_numpy(H).to(dev)
    nmf_algorithms_1D(dA, Wd, Hd, params=p).update()
    Wr, Hr = W.copy(), H.copy()
    orc.fro_hals_step_local(A, Wr, Hr, np.float32(EPS), W_update=True)
    assert _rel(Wd.cpu().numpy(), Wr) < 5e-5 and _rel(Hd.cpu().numpy(), Hr) < 5e-5


def _shapes_upd(seed, count):
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(count):
        k = int(rs.choice([1, 2, 3, 4, 5, 8, 12, 16, 17, 31, 32, 33, 40, 63, 64, 65, 96, 100, 127, 128]))
        r = int(rs.choice([1, 5, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 255, 256, 257, 1000, 1024, 2049, 4100, 70000]))
        pad = int(rs.choice([0, 0, 1, 3, 4, 8]))          # leading-dimension padding (elements)
        off = int(rs.choice([0, 0, 1, 2, 4]))             # start offset into the allocation (elements): alignment
        out.append((r, k, pad, off, int(rs.randint(1 << 30))))
    return out


def _view(rows, cols, pad, off, rs, dev, scale=1.0):
    """A rows x cols float32 view with leading dimension cols + pad, starting `off` elements into its buffer."""
    buf = torch.from_numpy((rs.rand(rows * (cols + pad) + off + 8) * scale).astype(np.float32)).to(dev)
    return torch.as_strided(buf, (rows, cols), (cols + pad, 1), off)


@pytest.mark.parametrize("r,k,pad,off,seed", _shapes_upd(_SEED + 7, max(60, _COUNT // 2)))
def test_random_shapes_update_kernels(r, k, pad, off, seed):
    """The stand-alone update / element-wise / statistics entry points (the ones multi-rank and 2D runs use) on views with
    padded leading dimensions and unaligned starts: dnmf_mu_update_w / _h (buffer-addressed MFMA kernels, vector and
    dword forms, interior and edge tiles), clamp / scale / KL updates (ew_kernel), column_err, the HALS W sweep."""
    from pydnmfk_amd.engine import HIP_OPS as ops, new_gram
    rs = np.random.RandomState(seed)
    dev = torch.device("cuda")
    G = new_gram(k, dev)
    g = rs.rand(k, k)
    G[:k, :k] = torch.from_numpy((g @ g.T + k).astype(np.float32)).to(dev)
    G64 = G[:k, :k].double().cpu().numpy()
    # W-type (r x k) and H-type (k x r)
    W, S = _view(r, k, pad, off, rs, dev), _view(r, k, pad, off, rs, dev, 20.0)
    W64, S64 = W.double().cpu().numpy(), S.double().cpu().numpy()
    Wd = W.clone() if pad == 0 and off == 0 else W        # views are updated in place
    ops.mu_update_w(Wd, S, G, EPS)
    assert _rel(Wd.cpu().numpy(), W64 * (S64 / (W64 @ G64 + EPS))) < 1e-5
    H, T = _view(k, r, pad, off, rs, dev), _view(k, r, pad, off, rs, dev, 20.0)
    H64, T64 = H.double().cpu().numpy(), T.double().cpu().numpy()
    ops.mu_update_h(H, T, G, EPS, True)
    assert _rel(H.cpu().numpy(), np.maximum(H64 * (T64 / (G64 @ H64 + EPS)), EPS)) < 1e-5
    # element-wise passes
    x = torch.from_numpy((rs.rand(k) + 0.5).astype(np.float32)).to(dev)
    x64 = x.double().cpu().numpy()
    W2, S2 = _view(r, k, pad, off, rs, dev), _view(r, k, pad, off, rs, dev)
    W264, S264 = W2.double().cpu().numpy(), S2.double().cpu().numpy()
    ops.kl_update_w(W2, S2, x, EPS)
    assert _rel(W2.cpu().numpy(), W264 * (S264 / (x64[None, :] + EPS))) < 1e-6
    H2, T2 = _view(k, r, pad, off, rs, dev), _view(k, r, pad, off, rs, dev)
    H264, T264 = H2.double().cpu().numpy(), T2.double().cpu().numpy()
    ops.kl_update_h(H2, T2, x, EPS, False)
    assert _rel(H2.cpu().numpy(), H264 * (T264 / (x64[:, None] + EPS))) < 1e-6
    W3 = _view(r, k, pad, off, rs, dev)
    W3[W3 < 0.3] = 0.0
    W364 = W3.double().cpu().numpy()
    ops.clamp_min(W3, EPS)
    assert np.array_equal(W3.cpu().numpy(), np.maximum(W364, EPS).astype(np.float32))
    ops.scale_cols_div(W3, x, EPS)
    assert _rel(W3.cpu().numpy(), np.maximum(W364, EPS) / (x64[None, :] + EPS)) < 1e-6
    H3 = _view(k, r, pad, off, rs, dev)
    H364 = H3.double().cpu().numpy()
    ops.scale_rows_mul(H3, x)
    assert _rel(H3.cpu().numpy(), H364 * x64[:, None]) < 1e-6
    # column_err sums on an r x n block
    n = int(rs.choice([4, 33, 128, 200]))
    A = _view(r, n, pad, off, rs, dev)
    Wc = torch.from_numpy(rs.rand(r, k).astype(np.float32)).to(dev)
    Hc = torch.from_numpy(rs.rand(k, n).astype(np.float32)).to(dev)
    num, den = ops.column_err_sums(A, Wc, Hc)
    A64 = A.double().cpu().numpy()
    R = A64 - Wc.double().cpu().numpy() @ Hc.double().cpu().numpy()
    assert _rel(num.cpu().numpy(), (R * R).sum(0)) < 2e-5 and _rel(den.cpu().numpy(), (A64 * A64).sum(0)) < 1e-6
    # HALS W sweep (persistent form) vs the float64 recursion
    if r >= 8:
        W4, AH = _view(r, k, pad, off, rs, dev), _view(r, k, pad, off, rs, dev, 300.0)
        Wr, A4 = W4.double().cpu().numpy(), AH.double().cpu().numpy()
        ops.hals_update_w(W4, AH, G, EPS)
        for kk in range(k):
            t = Wr[:, kk] * G64[kk, kk] + A4[:, kk] - Wr @ G64[:, kk]
            Wr[:, kk] = np.maximum(t, EPS)
            ss = np.linalg.norm(Wr[:, kk])
            if ss > 0:
                Wr[:, kk] /= ss
        assert _rel(W4.cpu().numpy(), Wr) < 1e-4


def _shapes_blocks(seed, count):
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(count):
        k = int(rs.choice([1, 3, 4, 8, 12, 16, 20, 32, 33, 64, 72, 128]))
        m = int(rs.choice([5, 32, 33, 96, 128, 130, 256, 300, 640, 1024, 1500, 4100]))
        nb = int(rs.choice([1, 2, 3, 4, 8]))
        nh = 32 * int(rs.choice([1, 2, 3, 4, 5, 8, 16]))
        out.append((m, nb, nh, k, int(rs.randint(1 << 30))))
    return out


@pytest.mark.parametrize("m,nb,nh,k,seed", _shapes_blocks(_SEED + 7, max(40, _COUNT // 4)))
def test_random_shapes_h_as_column_blocks(m, nb, nh, k, seed):
    """dnmf_aht_hblocks / dnmf_kl_uht_hblocks on random grids of column blocks: bit-identical (aht) / equal to rounding (kl_uht:
    the column splits may differ) to the same product with the assembled H, and right against float64."""
    from pydnmfk_amd.engine import HIP_OPS as ops
    rs = np.random.RandomState(seed)
    n = nb * nh
    A = rs.rand(m, n).astype(np.float32)
    A[rs.rand(m, n) < 0.1] = 0.0
    W = rs.rand(m, k).astype(np.float32)
    H = rs.rand(k, n).astype(np.float32)
    dev = torch.device("cuda")
    dA, dW, dH = (torch.from_numpy(x).to(dev) for x in (A, W, H))
    Hs = torch.from_numpy(np.ascontiguousarray(H.reshape(k, nb, nh).transpose(1, 0, 2))).to(dev)
    A64, W64, H64 = A.astype(np.float64), W.astype(np.float64), H.astype(np.float64)
    out = ops.aht_hblocks(dA, Hs, torch.full((m, k), 3.0, device=dev))
    assert torch.equal(out, ops.aht(dA, dH, torch.empty(m, k, device=dev)))
    assert _rel(out.cpu().numpy(), A64 @ H64.T) < 2e-6
    U = A64 / (W64 @ H64 + EPS)
    uht = ops.kl_uht_hblocks(dA, dW, Hs, EPS, torch.full((m, k), 3.0, device=dev))
    assert _rel(uht.cpu().numpy(), U @ H64.T) < 5e-6


def _shapes_large(seed, count):
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(count):
        k = int(rs.choice([2, 8, 16, 17, 32, 33, 64, 100, 128]))
        m = int(rs.choice([4097, 8192, 12345, 16384, 30000, 32768, 50001, 65536, 70000]))
        n = int(rs.choice([256, 1000, 1024, 2048, 3000, 4096, 4100, 8192]))
        out.append((m, n, k, int(rs.randint(0, 3)), int(rs.randint(1 << 30))))
    return out


@pytest.mark.parametrize("m,n,k,ldpad,seed", _shapes_large(_SEED + 11, 24))
def test_random_large_shapes(m, n, k, ldpad, seed):
    """The row-chunk / column-split plans at realistic sizes (W^T A with one or two waves per SIMD, the 16-wide kernels' slabs,
    split A H^T, many workgroup rounds, ragged last chunks, padded leading dimensions): the four big products and a whole
    MU/FRO + MU/KL step against float64 on the device (dist_nmf.py:705-751, 806-849)."""
    from pydnmfk_amd.engine import HIP_OPS as ops, new_gram
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(seed)
    lda = n + 4 * ldpad
    Abuf = torch.rand(m, lda, device=dev, generator=g)
    Abuf[torch.rand(m, lda, device=dev, generator=g) < 0.05] = 0.0
    A = Abuf[:, :n]
    W = torch.rand(m, k, device=dev, generator=g) + 0.05
    H = torch.rand(k, n, device=dev, generator=g) + 0.05
    A64, W64, H64 = A.double(), W.double(), H.double()

    def rel(x, ref):
        return float((x.double() - ref).norm() / ref.norm())

    assert rel(ops.aht(A, H, torch.full((m, k), 3.0, device=dev)), A64 @ H64.t()) < 2e-6
    assert rel(ops.wta(A, W, torch.full((k, n), 3.0, device=dev)), W64.t() @ A64) < 2e-6
    Ws, Hs = W.clone(), H.clone()
    ops.mu_fro_step(A, Ws, Hs, EPS, True, False)
    W_ref = W64 * ((A64 @ H64.t()) / (W64 @ (H64 @ H64.t()) + EPS))
    H_ref = H64 * ((W_ref.t() @ A64) / ((W_ref.t() @ W_ref) @ H64 + EPS))
    assert rel(Ws, W_ref) < 1e-5 and rel(Hs, H_ref) < 2e-5
    Ws, Hs = W.clone(), H.clone()
    ops.mu_kl_step(A, Ws, Hs, EPS, True, False)
    U = A64 / (W64 @ H64 + EPS)
    W_ref = W64 * (U @ H64.t()) / (H64.sum(1)[None, :] + EPS)
    U = A64 / (W_ref @ H64 + EPS)
    H_ref = H64 * (W_ref.t() @ U) / (W_ref.sum(0)[:, None] + EPS)
    assert rel(Ws, W_ref) < 1e-5 and rel(Hs, H_ref) < 2e-5
