"""Ranks 128 < k <= 256 (csrc/dnmf_wide.hip; the reference has no bound on k, dist_nmf.py:618-632): the contractions run as two
passes of the tuned kernels over A, the k x k products and the fused quotient / residual on plain MFMA kernels.  Every primitive
against a float64 evaluation on the device (the fp32 budgets of tests/test_gpu_kernels.py), whole fits against the oracle, the
batched fits bit-identical to single fits, and one NMFk sweep that ends at k = 140."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

EPS = float(np.finfo(np.float32).eps)


def rel(a, b):
    return float((a.double() - b).norm() / b.norm())


@pytest.mark.parametrize("m,n,k", [(1024, 512, 130), (777, 1000, 192), (2048, 1536, 256), (300, 260, 129), (4096, 384, 200)])
def test_wide_rank_primitives_match_float64(m, n, k):
    from pydnmfk_amd.engine import HIP_OPS as ops, kp
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(m + k)
    A = torch.rand(m, n, device=dev, generator=g)
    A[:, ::9] = 0.0
    W = torch.rand(m, k, device=dev, generator=g)
    H = torch.rand(k, n, device=dev, generator=g)
    Ad, Wd, Hd = A.double(), W.double(), H.double()
    KP = kp(k)
    assert KP == 256
    tol = 2e-6
    assert rel(ops.aht(A, H, torch.empty(m, k, device=dev)), Ad @ Hd.t()) < tol
    assert rel(ops.wta(A, W, torch.empty(k, n, device=dev)), Wd.t() @ Ad) < tol
    G = torch.full((KP, KP), 3.0, device=dev)
    ops.gram_hht(H, G)
    assert rel(G[:k, :k], Hd @ Hd.t()) < tol and (k == KP or (float(G[k:].abs().max()) == 0.0 and float(G[:, k:].abs().max()) == 0.0))
    G2 = torch.full((KP, KP), 3.0, device=dev)
    AtW = torch.empty(k, n, device=dev)
    ops.wta_gram(A, W, AtW, G2)
    assert rel(G2[:k, :k], Wd.t() @ Wd) < tol and (k == KP or float(G2[k:].abs().max()) == 0.0) and rel(AtW, Wd.t() @ Ad) < tol
    # updates (dist_nmf.py:731-732, :750-751) with exact inputs
    AH = (Ad @ Hd.t()).float()
    Gd = G.double()[:k, :k]
    Wn = W.clone(); ops.mu_update_w(Wn, AH, G, EPS)
    assert rel(Wn, Wd * (AH.double() / (Wd @ Gd + EPS))) < 5e-6
    Wf = W.clone(); ops.aht_update_w(A, H, G, Wf, EPS)
    assert rel(Wf, Wd * ((Ad @ Hd.t()) / (Wd @ Gd + EPS))) < 5e-6
    G2d = G2.double()[:k, :k]
    for clamp in (False, True):
        Hn = H.clone(); ops.mu_update_h(Hn, AtW, G2, EPS, clamp)
        ref = Hd * (AtW.double() / (G2d @ Hd + EPS))
        assert rel(Hn, torch.clamp(ref, min=EPS) if clamp else ref) < 5e-6
    # KL products (dist_nmf.py:806-810), residual, per-column error
    U = Ad / (Wd @ Hd + EPS)
    assert rel(ops.kl_uht(A, W, H, EPS, torch.empty(m, k, device=dev)), U @ Hd.t()) < 5e-6
    assert rel(ops.kl_wtu(A, W, H, EPS, torch.empty(k, n, device=dev)), Wd.t() @ U) < 5e-6
    Ws, Hs = W / k, H                                                   # a residual that is not dominated by W H
    R = Ad - Ws.double() @ Hs.double()
    assert abs(float(ops.resid_sqnorm(A, Ws, Hs)) / float((R * R).sum()) - 1) < 1e-5
    num, den = ops.column_err_sums(A, Ws, Hs)
    assert rel(num, (R * R).sum(0)) < 1e-5 and rel(den[den > 0], (Ad * Ad).sum(0)[den > 0]) < 1e-6
    x = torch.empty(k, device=dev)
    assert rel(ops.colsum(W, x), Wd.sum(0)) < 1e-6
    # HALS sweeps against the recursion in float64 from the same inputs (one sweep: no trajectory amplification)
    Wh = W.clone(); ops.hals_update_w(Wh, AH, G, EPS)
    Wr = Wd.clone()
    AHd = AH.double()
    for kk in range(k):
        Wr[:, kk] = torch.clamp(Wr[:, kk] * Gd[kk, kk] + AHd[:, kk] - Wr @ Gd[:, kk], min=EPS)
        nrm = Wr[:, kk].norm()
        if nrm > 0:
            Wr[:, kk] /= nrm
    assert rel(Wh, Wr) < 5e-4
    Hh = H.clone(); ops.hals_update_h(Hh, AtW, G2, EPS)
    Hr = Hd.clone()
    for kk in range(k):
        Hr[kk] = torch.clamp(Hr[kk] + AtW.double()[kk] - G2d[kk] @ Hr, min=EPS)
    assert rel(Hh, Hr) < 5e-4


def _args(k, itr, norm, method="mu", **kw):
    from pydnmfk_amd.dist_comm import MPI_comm
    from pydnmfk_amd.utils import parse
    comms = MPI_comm(None, 1, 1)
    args = parse()
    args.comm1, args.comm, args.p_r, args.p_c, args.k = comms.comm, comms, 1, 1, k
    args.row_comm, args.col_comm = comms.cart_1d_row(), comms.cart_1d_column()
    args.itr, args.init, args.verbose, args.prune = itr, "rand", False, False
    args.norm, args.method, args.W_update = norm, method, True
    for key, v in kw.items():
        setattr(args, key, v)
    return args


@pytest.mark.parametrize("k,norm,method", [(130, "fro", "mu"), (192, "kl", "mu"), (140, "fro", "hals")])
def test_wide_rank_fit_matches_oracle(k, norm, method):
    """whole fits (one library call: dnmf_*_fit) and the per-step loop against the numpy oracle on a low-rank-plus-noise matrix"""
    from oracle import nmf_oracle as orc
    from pydnmfk_amd.pyDNMF import PyNMF
    rs = np.random.RandomState(100)
    m, n, itr = 640, 512, 6
    A = np.abs(rs.rand(m, 20) @ rs.rand(20, n) + 0.01 * rs.randn(m, n)).astype(np.float32)
    W0, H0 = rs.rand(m, k).astype(np.float32), rs.rand(k, n).astype(np.float32)
    Wr, Hr, err_r = orc.fit_single(A, W0, H0, itr, norm=norm, method=method)
    tol = 2e-3 if method == "hals" else 1e-4
    for loop in ("native", "python"):
        W, H, err = PyNMF(A, factors=[W0, H0], params=_args(k, itr, norm, method, fit_loop=loop)).fit()
        assert np.linalg.norm(W - Wr) / np.linalg.norm(Wr) < tol and np.linalg.norm(H - Hr) / np.linalg.norm(Hr) < tol, loop
        assert abs(err - err_r) < 1e-5 * max(1.0, err_r), loop


@pytest.mark.parametrize("norm,method", [("fro", "mu"), ("kl", "mu"), ("fro", "hals")])
def test_wide_rank_batched_fit_is_bit_identical(norm, method):
    from pydnmfk_amd.pyDNMF import PyNMF
    m, n, k, itr, B = 512, 384, 150, 5, 3
    probs = []
    for b in range(B):
        g = torch.Generator(device="cuda").manual_seed(50 + b)
        probs.append((torch.rand(m, n, device="cuda", generator=g) + 0.01, torch.rand(m, k, device="cuda", generator=g),
                      torch.rand(k, n, device="cuda", generator=g)))
    single = [PyNMF(A, factors=[W0, H0], params=_args(k, itr, norm, method)).fit() for A, W0, H0 in probs]
    batched = PyNMF.fit_batch([PyNMF(A, factors=[W0, H0], params=_args(k, itr, norm, method)) for A, W0, H0 in probs])
    for b in range(B):
        assert torch.equal(batched[b][0], single[b][0]) and torch.equal(batched[b][1], single[b][1]), b
        assert abs(batched[b][2] - single[b][2]) <= 1e-9 * max(1.0, abs(single[b][2])), b


def test_nmfk_sweep_up_to_k_140(tmp_path):
    """an NMFk sweep whose range ends beyond the tuned kernels' rank (end_k = 140): perturbation fits, clustering, regression fit
    and column errors all run at k = 138 .. 140"""
    from pydnmfk_amd.dist_comm import MPI_comm
    from pydnmfk_amd.pyDNMFk import PyNMFk
    from pydnmfk_amd.utils import parse
    rs = np.random.RandomState(3)
    A = torch.from_numpy((rs.rand(768, 6) @ rs.rand(6, 512) + 0.01 * rs.rand(768, 512)).astype(np.float32)).cuda()
    comms = MPI_comm(None, 1, 1)
    args = parse()
    args.comm1, args.comm, args.p_r, args.p_c = comms.comm, comms, 1, 1
    args.row_comm, args.col_comm = comms.cart_1d_row(), comms.cart_1d_column()
    args.fpath, args.fname, args.ftype = str(tmp_path) + "/", "wide", "npy"
    args.start_k, args.end_k, args.step_k, args.sill_thr, args.itr, args.init = 138, 140, 1, 0.8, 40, "rand"
    args.noise_var, args.verbose, args.norm, args.method, args.checkpoint = 0.03, False, "fro", "mu", False
    args.prune, args.perturbations, args.rng = False, 3, "device"
    args.results_path = str(tmp_path) + "/results/"
    nm = PyNMFk(A, factors=None, params=args)
    nopt = nm.fit()
    assert 138 <= nopt <= 140 and sorted(nm.stats) == [138, 139, 140]
    for k, st in nm.stats.items():
        assert np.isfinite(st["avgErr"]) and st["avgErr"] < 0.3 and np.asarray(st["L_err"]).shape == (512,)
        assert np.all(np.isfinite(np.asarray(st["clusterSilhouetteCoefficients"])))
