"""bf16x6: the contractions A H^T and W^T A as six bf16 piece products per fp32 product (csrc/dnmf_split.h).  The claim
tested here is "fp32 grade": against a float64 product the split path is as close as the fp32-MFMA path, the updates built
on it follow the fp32 ones to rounding level, the goldens of the fp32 path pass at the same tolerances, and shapes without a
split kernel run the fp32 kernels bit for bit."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
from tests.conftest import long_param  # noqa: E402
pytestmark = pytest.mark.gpu

EPS = float(np.finfo(np.float32).eps)


def _ops():
    from pydnmfk_amd.engine import HIP_OPS, HIP_OPS_BF16X6, new_gram
    return HIP_OPS, HIP_OPS_BF16X6, new_gram


def _rand(m, n, k, seed, scale=None):
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(seed)
    A = torch.rand(m, n, device=dev, generator=g)
    W = torch.rand(m, k, device=dev, generator=g)
    H = torch.rand(k, n, device=dev, generator=g)
    if scale == "wide":           # entries over 12 decades, both signs of the exponent: the pieces of tiny and huge values
        A = A * torch.exp(14.0 * (torch.rand(m, n, device=dev, generator=g) - 0.5))
        H = H * torch.exp(14.0 * (torch.rand(k, n, device=dev, generator=g) - 0.5))
        W = W * torch.exp(14.0 * (torch.rand(m, k, device=dev, generator=g) - 0.5))
    return A, W, H


@pytest.mark.parametrize("m,n,k,scale", [(4096, 1024, 64, None), (5000, 2048, 40, None), (100, 128, 33, None),
                                         (777, 384, 64, None), (4096, 1024, 64, "wide"), (16384, 8192, 64, None),
                                         (4096, 1024, 128, None), (5000, 2048, 100, None), (300, 256, 65, "wide")])
def test_products_are_fp32_grade(m, n, k, scale):
    """max and rms relative error against float64: the split path is within 1.25x of the fp32-MFMA path (it is usually the
    closer one: 16 products per accumulator update instead of 2), and both are at rounding level."""
    f32, x6, _ = _ops()
    A, W, H = _rand(m, n, k, 11, scale)
    dev = A.device
    AH0, AH1 = torch.empty(m, k, device=dev), torch.empty(m, k, device=dev)
    AtW0, AtW1 = torch.empty(k, n, device=dev), torch.empty(k, n, device=dev)
    f32.aht(A, H, AH0); x6.aht(A, H, AH1)
    f32.wta(A, W, AtW0); x6.wta(A, W, AtW1)
    ref_ah = A.double() @ H.double().t()
    ref_wa = W.double().t() @ A.double()
    for got0, got1, ref, depth in ((AH0, AH1, ref_ah, n), (AtW0, AtW1, ref_wa, m)):
        e0 = ((got0.double() - ref) / ref).abs()
        e1 = ((got1.double() - ref) / ref).abs()
        assert float(e1.max()) <= max(1.25 * float(e0.max()), 8 * EPS)
        assert float(e1.pow(2).mean().sqrt()) <= max(1.25 * float(e0.pow(2).mean().sqrt()), 2 * EPS)
        assert float(e1.max()) <= 4 * EPS * np.sqrt(depth)          # rounding level for a sum of `depth` positive terms


@pytest.mark.parametrize("m,n,k", [(4096, 1024, 64), (5000, 2048, 40), (300, 256, 50), (2048, 1024, 128), (1000, 384, 97)])
def test_fused_w_update_and_step_follow_fp32(m, n, k):
    f32, x6, new_gram = _ops()
    A, W, H = _rand(m, n, k, 5)
    G = f32.gram_hht(H, new_gram(k, A.device))
    Wa, Wb = W.clone(), W.clone()
    f32.aht_update_w(A, H, G, Wa, EPS)
    x6.aht_update_w(A, H, G, Wb, EPS)
    assert float((Wa - Wb).abs().max() / Wa.abs().max()) < 5e-6       # both are ~3e-6 from the float64 result at this depth
    Wa, Ha, Wb, Hb = W.clone(), H.clone(), W.clone(), H.clone()
    for i in range(10):
        f32.mu_fro_step(A, Wa, Ha, EPS, True, i % 10 == 0)
        x6.mu_fro_step(A, Wb, Hb, EPS, True, i % 10 == 0)
    assert float((Wa - Wb).norm() / Wa.norm()) < 1e-5 and float((Ha - Hb).norm() / Ha.norm()) < 1e-5
    assert torch.isfinite(Wb).all() and torch.isfinite(Hb).all()


def test_step_matches_oracle_at_fp32_tolerance():
    """The fp32 path's own parity bar (tests/test_gpu_parity.py: one MU step against the numpy restatement, 1e-5)."""
    from oracle import nmf_oracle as orc
    _, x6, _ = _ops()
    rs = np.random.RandomState(3)
    m, n, k = 2048, 512, 48
    A = np.abs(rs.rand(m, k) @ rs.rand(k, n) + 0.01 * rs.randn(m, n)).astype(np.float32)
    W0, H0 = rs.rand(m, k).astype(np.float32), rs.rand(k, n).astype(np.float32)
    Wr, Hr = orc.fro_mu_step_local(A, W0.copy(), H0.copy(), np.float32(EPS))
    dev = torch.device("cuda")
    Ad, Wd, Hd = (torch.from_numpy(v).to(dev) for v in (A, W0, H0))
    x6.mu_fro_step(Ad, Wd, Hd, EPS, True, False)
    assert np.linalg.norm(Wd.cpu().numpy() - Wr) / np.linalg.norm(Wr) < 1e-5
    assert np.linalg.norm(Hd.cpu().numpy() - Hr) / np.linalg.norm(Hr) < 1e-5


@pytest.mark.parametrize("m,n,k", [(1000, 700, 40), (4096, 1024, 16), (2048, 520, 128), (512, 1000, 64)])
def test_shapes_without_a_split_kernel_run_the_fp32_kernels(m, n, k):
    f32, x6, _ = _ops()
    A, W, H = _rand(m, n, k, 9)
    Wa, Ha, Wb, Hb = W.clone(), H.clone(), W.clone(), H.clone()
    for _ in range(3):
        f32.mu_fro_step(A, Wa, Ha, EPS)
        x6.mu_fro_step(A, Wb, Hb, EPS)
    assert torch.equal(Wa, Wb) and torch.equal(Ha, Hb)
    out0, out1 = torch.empty(k, n, device=A.device), torch.empty(k, n, device=A.device)
    assert torch.equal(f32.wta(A, W, out0), x6.wta(A, W, out1))


def test_split_steps_are_bitwise_reproducible():
    _, x6, _ = _ops()
    A, W, H = _rand(8192, 1024, 64, 21)
    outs = []
    for _ in range(2):
        Wc, Hc = W.clone(), H.clone()
        for i in range(6):
            x6.mu_fro_step(A, Wc, Hc, EPS, True, i % 10 == 0)
        outs.append((Wc, Hc))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_views_with_a_row_pitch_and_zero_rows():
    """A as a column slice of a wider matrix (lda > n) and all-zero rows / columns (0 * pieces stay 0, no NaN)."""
    f32, x6, _ = _ops()
    m, n, k = 1024, 512, 64
    big, W, H = _rand(m, 768, k, 4)
    A = big[:, 128:128 + n]
    A[17] = 0
    A[:, 5] = 0
    H = H[:, :n].contiguous()
    o0, o1 = torch.empty(m, k, device=A.device), torch.empty(m, k, device=A.device)
    f32.aht(A, H, o0); x6.aht(A, H, o1)
    assert float(o1[17].abs().max()) == 0.0
    assert float((o0 - o1).abs().max() / o0.abs().max()) < 5e-6
    t0, t1 = torch.empty(k, n, device=A.device), torch.empty(k, n, device=A.device)
    f32.wta(A, W, t0); x6.wta(A, W, t1)
    assert float(t1[:, 5].abs().max()) == 0.0
    assert float((t0 - t1).abs().max() / t0.abs().max()) < 5e-6


def test_pynmf_with_gemm_option_matches_default():
    """params.gemm = 'bf16x6' through the reference-shaped entry point: same factors as the default to rounding level."""
    from pydnmfk_amd.pyDNMF import PyNMF
    from pydnmfk_amd.utils import parse
    from pydnmfk_amd.dist_comm import MPI_comm
    rs = np.random.RandomState(0)
    m, n, k = 1024, 512, 40
    A = np.abs(rs.rand(m, k) @ rs.rand(k, n)).astype(np.float32)
    W0, H0 = rs.rand(m, k).astype(np.float32), rs.rand(k, n).astype(np.float32)
    res = []
    for gemm in ("fp32", "bf16x6"):
        comms = MPI_comm(None, 1, 1)
        p = parse()
        p.comm1, p.comm, p.p_r, p.p_c, p.k = comms.comm, comms, 1, 1, k
        p.row_comm, p.col_comm = comms.cart_1d_row(), comms.cart_1d_column()
        p.norm, p.method, p.itr, p.init, p.verbose, p.prune, p.W_update, p.gemm = "fro", "mu", 30, "rand", False, False, True, gemm
        W, H, err = PyNMF(A, factors=[W0.copy(), H0.copy()], params=p).fit()
        res.append((np.asarray(W), np.asarray(H), err))
    assert np.linalg.norm(res[0][0] - res[1][0]) / np.linalg.norm(res[0][0]) < 1e-4
    assert np.linalg.norm(res[0][1] - res[1][1]) / np.linalg.norm(res[0][1]) < 1e-4
    with pytest.raises(ValueError):
        from pydnmfk_amd.engine import ops_for
        bad = parse(); bad.gemm = "fp8"
        ops_for(bad)


# ---------------------------------------------------------------------------------------------- KL products
def _kl_ref(A, W, H, eps):
    U = A.double() / (W.double() @ H.double() + eps)
    return U @ H.double().t(), W.double().t() @ U


@pytest.mark.parametrize("m,n,k", [(1024, 512, 16), (1000, 384, 40), (333, 128, 5), (4096, 1024, 64), (130, 256, 1),
                                   (2048, 2048, 32), (5000, 1280, 33), (1000, 384, 100), (4096, 1024, 128), (333, 256, 65)])
def test_kl_products_are_fp32_grade(m, n, k):
    """U H^T and W^T U with U = A / (W H + eps): the split path against float64, next to the fp32-MFMA path."""
    f32, x6, _ = _ops()
    A, W, H = _rand(m, n, k, 13)
    dev = A.device
    ref_u, ref_t = _kl_ref(A, W, H, EPS)
    for name, ref, shape in (("kl_uht", ref_u, (m, k)), ("kl_wtu", ref_t, (k, n))):
        o0, o1 = torch.empty(*shape, device=dev), torch.empty(*shape, device=dev)
        getattr(f32, name)(A, W, H, EPS, o0)
        getattr(x6, name)(A, W, H, EPS, o1)
        e0 = ((o0.double() - ref) / ref).abs()
        e1 = ((o1.double() - ref) / ref).abs()
        assert float(e1.max()) <= max(1.25 * float(e0.max()), 8 * EPS), name
        assert float(e1.pow(2).mean().sqrt()) <= max(1.25 * float(e0.pow(2).mean().sqrt()), 2 * EPS), name


@pytest.mark.parametrize("m,n,k", [(1000, 384, 40), (2048, 512, 8), (4096, 1024, 64), (2048, 768, 128)])
def test_kl_step_follows_fp32_and_oracle(m, n, k):
    from oracle import nmf_oracle as orc
    f32, x6, _ = _ops()
    rs = np.random.RandomState(8)
    A = np.abs(rs.rand(m, k) @ rs.rand(k, n) + 0.01 * rs.randn(m, n)).astype(np.float32)
    W0, H0 = rs.rand(m, k).astype(np.float32), rs.rand(k, n).astype(np.float32)
    Wr, Hr = orc.kl_mu_step_local(A, W0.copy(), H0.copy(), np.float32(EPS))
    dev = torch.device("cuda")
    Ad = torch.from_numpy(A).to(dev)
    Wb, Hb = torch.from_numpy(W0).to(dev), torch.from_numpy(H0).to(dev)
    x6.mu_kl_step(Ad, Wb, Hb, EPS, True, False)
    assert np.linalg.norm(Wb.cpu().numpy() - Wr) / np.linalg.norm(Wr) < 1e-5      # the fp32 path's bar for one step
    assert np.linalg.norm(Hb.cpu().numpy() - Hr) / np.linalg.norm(Hr) < 1e-5
    Wa, Ha = torch.from_numpy(W0).to(dev), torch.from_numpy(H0).to(dev)
    Wb, Hb = Wa.clone(), Ha.clone()
    for i in range(10):
        f32.mu_kl_step(Ad, Wa, Ha, EPS, True, i % 10 == 0)
        x6.mu_kl_step(Ad, Wb, Hb, EPS, True, i % 10 == 0)
    assert float((Wa - Wb).norm() / Wa.norm()) < 1e-5 and float((Ha - Hb).norm() / Ha.norm()) < 1e-5


def test_kl_zero_rows_and_views():
    """All-zero rows / columns of A give exact zeros (U = 0 / (S + eps)), A as a column view of a wider matrix."""
    f32, x6, _ = _ops()
    m, n, k = 777, 256, 12
    big, W, H = _rand(m, 512, k, 17)
    A = big[:, 128:128 + n]
    A[5] = 0
    A[:, 9] = 0
    H = H[:, :n].contiguous()
    u0, u1 = torch.empty(m, k, device=A.device), torch.empty(m, k, device=A.device)
    f32.kl_uht(A, W, H, EPS, u0); x6.kl_uht(A, W, H, EPS, u1)
    assert float(u1[5].abs().max()) == 0.0 and float((u0 - u1).abs().max() / u0.abs().max()) < 5e-6
    t0, t1 = torch.empty(k, n, device=A.device), torch.empty(k, n, device=A.device)
    f32.kl_wtu(A, W, H, EPS, t0); x6.kl_wtu(A, W, H, EPS, t1)
    assert float(t1[:, 9].abs().max()) == 0.0 and float((t0 - t1).abs().max() / t0.abs().max()) < 5e-6


@pytest.mark.parametrize("m,n,k", [(1000, 700, 40), (2048, 520, 128)])
def test_kl_shapes_without_a_split_kernel_run_the_fp32_kernels(m, n, k):
    f32, x6, _ = _ops()
    A, W, H = _rand(m, n, k, 19)
    Wa, Ha, Wb, Hb = W.clone(), H.clone(), W.clone(), H.clone()
    for _ in range(2):
        f32.mu_kl_step(A, Wa, Ha, EPS)
        x6.mu_kl_step(A, Wb, Hb, EPS)
    assert torch.equal(Wa, Wb) and torch.equal(Ha, Hb)


def test_kl_split_steps_are_bitwise_reproducible():
    _, x6, _ = _ops()
    A, W, H = _rand(4096, 1024, 20, 23)
    outs = []
    for _ in range(2):
        Wc, Hc = W.clone(), H.clone()
        for i in range(5):
            x6.mu_kl_step(A, Wc, Hc, EPS, True, i % 10 == 0)
        outs.append((Wc, Hc))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("seed", [0, 1] + [long_param(x) for x in range(2, 6)])
def test_random_shapes_split_vs_fp32(seed):
    """Ragged row counts, every rank tile, column views with a pitch: each split product against its fp32 twin (both within
    rounding of the exact result, so they agree to a few 1e-6 of the largest entry)."""
    f32, x6, new_gram = _ops()
    rs = np.random.RandomState(1000 + seed)
    dev = torch.device("cuda")
    for _ in range(12):
        m = int(rs.choice([1, 7, 31, 32, 33, 100, 129, 500, 1023, 2500]))
        n = 128 * int(rs.randint(1, 9))
        k = int(rs.choice([1, 2, 5, 16, 17, 32, 33, 40, 64, 65, 96, 127, 128]))
        pitch = n + 4 * int(rs.randint(0, 3)) * int(rs.randint(0, 2))
        g = torch.Generator(device="cuda").manual_seed(int(rs.randint(1 << 30)))
        big = torch.rand(m, pitch, device=dev, generator=g) + 0.01
        A = big[:, :n]
        W = torch.rand(m, k, device=dev, generator=g) + 0.01
        H = torch.rand(k, n, device=dev, generator=g) + 0.01
        tag = (m, n, k, pitch)
        for name, shape in (("aht", (m, k)), ("wta", (k, n))):
            o0, o1 = torch.empty(*shape, device=dev), torch.empty(*shape, device=dev)
            args = (A, H) if name == "aht" else (A, W)
            getattr(f32, name)(*args, o0)
            getattr(x6, name)(*args, o1)
            assert float((o0 - o1).abs().max() / o0.abs().max()) < 5e-6, (name, tag)
        for name, shape in (("kl_uht", (m, k)), ("kl_wtu", (k, n))):
            o0, o1 = torch.empty(*shape, device=dev), torch.empty(*shape, device=dev)
            getattr(f32, name)(A, W, H, EPS, o0)
            getattr(x6, name)(A, W, H, EPS, o1)
            assert float((o0 - o1).abs().max() / o0.abs().max()) < 5e-6, (name, tag)
        Wa, Ha, Wb, Hb = W.clone(), H.clone(), W.clone(), H.clone()
        f32.mu_fro_step(A, Wa, Ha, EPS); x6.mu_fro_step(A, Wb, Hb, EPS)
        assert float((Wa - Wb).abs().max() / Wa.abs().max()) < 2e-5 and float((Ha - Hb).abs().max() / Ha.abs().max()) < 2e-5, tag
        Wa, Ha, Wb, Hb = W.clone(), H.clone(), W.clone(), H.clone()
        f32.mu_kl_step(A, Wa, Ha, EPS); x6.mu_kl_step(A, Wb, Hb, EPS)
        assert float((Wa - Wb).abs().max() / Wa.abs().max()) < 2e-5 and float((Ha - Hb).abs().max() / Ha.abs().max()) < 2e-5, tag


# ---------------------------------------------------------------------------------------------- bf16-stored A: three products
@pytest.mark.parametrize("m,n,k", [(4096, 1024, 64), (5000, 2048, 100), (333, 256, 128), (100, 128, 33), (2048, 512, 40),
                                   (4096, 1024, 32), (1000, 384, 17), (2048, 256, 24)])
def test_bf16_stored_a_products_are_fp32_grade(m, n, k):
    """A stored as bfloat16 is its own single piece: A times the three pieces of the fp32 factor.  Reference = float64 product of
    float(A); the fp32-MFMA bf16a kernels are the yardstick."""
    f32, x6, _ = _ops()
    A, W, H = _rand(m, n, k, 31)
    Ab = A.to(torch.bfloat16)
    dev = A.device
    ref_ah = Ab.double() @ H.double().t()
    ref_wa = W.double().t() @ Ab.double()
    for name, ref, args, shape in (("aht", ref_ah, (Ab, H), (m, k)), ("wta", ref_wa, (Ab, W), (k, n))):
        o0, o1 = torch.empty(*shape, device=dev), torch.empty(*shape, device=dev)
        getattr(f32, name)(*args, o0)
        getattr(x6, name)(*args, o1)
        e0 = ((o0.double() - ref) / ref).abs()
        e1 = ((o1.double() - ref) / ref).abs()
        assert float(e1.max()) <= max(1.25 * float(e0.max()), 8 * EPS), name
        assert float(e1.pow(2).mean().sqrt()) <= max(1.25 * float(e0.pow(2).mean().sqrt()), 2 * EPS), name
    Wa, Ha, Wb, Hb = W.clone(), H.clone(), W.clone(), H.clone()
    for i in range(6):
        f32.mu_fro_step(Ab, Wa, Ha, EPS, True, i % 10 == 0)
        x6.mu_fro_step(Ab, Wb, Hb, EPS, True, i % 10 == 0)
    assert float((Wa - Wb).norm() / Wa.norm()) < 1e-5 and float((Ha - Hb).norm() / Ha.norm()) < 1e-5


def test_bf16_stored_a_fallback_shapes_and_reproducibility():
    f32, x6, _ = _ops()
    A, W, H = _rand(1000, 700, 40, 37)                 # n % 128 != 0: the fp32-MFMA bf16a kernels, bit for bit
    Ab = A.to(torch.bfloat16)
    Wa, Ha, Wb, Hb = W.clone(), H.clone(), W.clone(), H.clone()
    f32.mu_fro_step(Ab, Wa, Ha, EPS); x6.mu_fro_step(Ab, Wb, Hb, EPS)
    assert torch.equal(Wa, Wb) and torch.equal(Ha, Hb)
    A, W, H = _rand(4096, 1024, 64, 41)
    Ab = A.to(torch.bfloat16)
    outs = []
    for _ in range(2):
        Wc, Hc = W.clone(), H.clone()
        for i in range(5):
            x6.mu_fro_step(Ab, Wc, Hc, EPS, True, False)
        outs.append((Wc, Hc))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_wide_pitch_views_fall_back():
    """A column view whose row pitch exceeds 4 n is outside the split kernels' 32-bit tile addressing: forwarded to the fp32
    kernels (bit-identical results), for the Frobenius and the KL entry points alike."""
    f32, x6, _ = _ops()
    big, W, H = _rand(2048, 1024, 40, 43)
    A = big[:, :128]                                    # lda = 1024 = 8 n
    H = H[:, :128].contiguous()
    for name, args, shape in (("aht", (A, H), (2048, 40)), ("wta", (A, W), (40, 128))):
        o0, o1 = torch.empty(*shape, device=A.device), torch.empty(*shape, device=A.device)
        getattr(f32, name)(*args, o0); getattr(x6, name)(*args, o1)
        assert torch.equal(o0, o1), name
    for name, shape in (("kl_uht", (2048, 40)), ("kl_wtu", (40, 128))):
        o0, o1 = torch.empty(*shape, device=A.device), torch.empty(*shape, device=A.device)
        getattr(f32, name)(A, W, H, EPS, o0); getattr(x6, name)(A, W, H, EPS, o1)
        assert torch.equal(o0, o1), name
