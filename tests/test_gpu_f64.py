"""The float64 path (csrc/dnmf_f64.hip, engine.HipOpsF64): the reference computes in the dtype of A_ij (pyDNMF.py:68) and its
own tests feed float64 arrays (tests/test_dist_nmf_1d.py:14-20).

  * every primitive against a float64 evaluation with torch on the device (rel-Frobenius <= 1e-13; the sums differ only in order);
  * the nine `t24x12_*_float64` goldens captured from the unmodified reference: one update step <= 1e-10, fits <= 1e-8 on W and H,
    |recon_err difference| <= 1e-9 -- 1 x 1 directly, the grids with one process per rank on the one GPU (gloo transport);
  * the reference's own test recipe (tests/test_dist_nmf_1d.py:14-46: exact rank-2 float64 data, rel_error < 1e-3 / 1e-1).
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from tests._golden import case_names, load_case, rel_fro  # noqa: E402

F64 = [c for c in case_names() if c.endswith("float64")]
EPS = float(np.finfo(np.float64).eps)


def rel(a, b):
    return float((a - b).norm() / b.norm())


@pytest.fixture(scope="module")
def ops():
    from pydnmfk_amd.engine import HIP_OPS_F64
    return HIP_OPS_F64


@pytest.mark.parametrize("m,n,k", [(24, 12, 2), (300, 200, 7), (1024, 512, 16), (1000, 777, 33), (2048, 1024, 64), (513, 1100, 128),
                                   (20000, 96, 5), (64, 20000, 40)])
def test_f64_primitives_match_torch(ops, m, n, k):
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(m + n + k)
    A = torch.rand(m, n, dtype=torch.float64, device=dev, generator=g)
    A[:, ::5] = 0.0
    W = torch.rand(m, k, dtype=torch.float64, device=dev, generator=g)
    H = torch.rand(k, n, dtype=torch.float64, device=dev, generator=g)
    tol = 1e-13
    out = torch.full((m, k), 7.0, dtype=torch.float64, device=dev)
    assert rel(ops.aht(A, H, out), A @ H.t()) < tol
    out = torch.full((k, n), 7.0, dtype=torch.float64, device=dev)
    assert rel(ops.wta(A, W, out), W.t() @ A) < tol
    G = torch.full((k, k), 7.0, dtype=torch.float64, device=dev)
    assert rel(ops.gram_hht(H, G), H @ H.t()) < tol
    G2 = torch.full((k, k), 7.0, dtype=torch.float64, device=dev)
    assert rel(ops.gram_wtw(W, G2), W.t() @ W) < tol
    # Gram buffers with a pitch (the choreography hands KP x KP views): only the k x k block is written
    Gp = torch.full((k + 5, k + 5), 7.0, dtype=torch.float64, device=dev)
    ops.gram_wtw(W, Gp[:k, :k])
    assert rel(Gp[:k, :k], W.t() @ W) < tol and float(Gp[k:].min()) == 7.0 and float(Gp[:, k:].min()) == 7.0
    # multiplicative updates (dist_nmf.py:731-732, :750-751)
    AH = A @ H.t()
    Wn = W.clone()
    ops.mu_update_w(Wn, AH, G, EPS)
    assert rel(Wn, W * (AH / (W @ G + EPS))) < tol
    AtW = W.t() @ A
    for clamp in (False, True):
        Hn = H.clone()
        ops.mu_update_h(Hn, AtW, G2, EPS, clamp)
        ref = H * (AtW / (G2 @ H + EPS))
        assert rel(Hn, torch.clamp(ref, min=EPS) if clamp else ref) < tol
    Wf = W.clone()
    ops.aht_update_w(A, H, G, Wf, EPS)
    assert rel(Wf, W * (AH / (W @ G + EPS))) < tol
    # KL products (dist_nmf.py:806-810) and the element-wise KL updates
    U = A / (W @ H + EPS)
    o1 = torch.empty(m, k, dtype=torch.float64, device=dev)
    assert rel(ops.kl_uht(A, W, H, EPS, o1), U @ H.t()) < tol
    o2 = torch.empty(k, n, dtype=torch.float64, device=dev)
    assert rel(ops.kl_wtu(A, W, H, EPS, o2), W.t() @ U) < tol
    x = torch.empty(k, dtype=torch.float64, device=dev)
    assert rel(ops.rowsum(H, x), H.sum(1)) < tol
    x2 = torch.empty(k, dtype=torch.float64, device=dev)
    assert rel(ops.colsum(W, x2), W.sum(0)) < tol
    Wk = W.clone(); ops.kl_update_w(Wk, o1, x, EPS)
    assert rel(Wk, W * (o1 / (x + EPS))) < tol
    Hk = H.clone(); ops.kl_update_h(Hk, o2, x2, EPS, True)
    assert rel(Hk, torch.clamp(H * (o2 / (x2[:, None] + EPS)), min=EPS)) < tol
    # fit helpers
    Wc = (W - 0.5).clone(); ops.clamp_min(Wc, EPS)
    assert torch.equal(Wc, torch.clamp(W - 0.5, min=EPS))
    Ws = W.clone(); ops.scale_cols_div(Ws, x2, EPS)
    assert rel(Ws, W / (x2 + EPS)) < tol
    Hs = H.clone(); ops.scale_rows_mul(Hs, x2)
    assert rel(Hs, H * x2[:, None]) < tol
    assert abs(float(ops.sqnorm(A)) / float((A * A).sum()) - 1) < tol
    R = A - W @ H
    assert abs(float(ops.resid_sqnorm(A, W, H)) / float((R * R).sum()) - 1) < tol
    num, den = ops.column_err_sums(A, W, H)
    assert rel(num, (R * R).sum(0)) < tol and rel(den, (A * A).sum(0)) < tol
    # HALS sweeps (dist_nmf.py:884-891, :905-909) against the recursion in torch
    Wh, Wr = W.clone(), W.clone()
    ops.hals_update_w(Wh, AH, G, EPS)
    for kk in range(k):
        Wr[:, kk] = torch.clamp(Wr[:, kk] * G[kk, kk] + AH[:, kk] - Wr @ G[:, kk], min=EPS)
        nrm = Wr[:, kk].norm()
        if nrm > 0:
            Wr[:, kk] /= nrm
    assert rel(Wh, Wr) < 1e-11
    Hh, Hr = H.clone(), H.clone()
    ops.hals_update_h(Hh, AtW, G2, EPS)
    for kk in range(k):
        Hr[kk] = torch.clamp(Hr[kk] + AtW[kk] - G2[kk] @ Hr, min=EPS)
    assert rel(Hh, Hr) < 1e-11


@pytest.mark.parametrize("m,n,k", [(33, 20, 1), (500, 77, 7), (1000, 2048, 16), (700, 1500, 17), (4100, 530, 32), (2500, 1030, 33),
                                   (9000, 260, 48), (3000, 1200, 64), (70000, 96, 64), (40000, 130, 12), (1500, 300, 65)])
def test_f64_fused_kl_products(ops, m, n, k):
    """The KL products that keep the quotient in registers (k <= 64; csrc/dnmf_f64_kl.h) against torch float64 on the same operands:
    every tile count of k, ragged rows / columns, the column-split form of a short A (partial slabs + ordered reduction), both row-tile
    counts per wave; k = 65 takes the image path behind the same entry point.  Padded outputs: only the product block is written."""
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(11 * m + 3 * n + k)
    A = torch.rand(m, n, dtype=torch.float64, device=dev, generator=g)
    A[A < 0.2] = 0.0                                                  # (sparse-ish counts, as KL data are)
    W = torch.rand(m, k, dtype=torch.float64, device=dev, generator=g)
    H = torch.rand(k, n, dtype=torch.float64, device=dev, generator=g)
    U = A / (W @ H + EPS)
    S1 = torch.full((m, k + 3), 7.0, dtype=torch.float64, device=dev)
    ops.kl_uht(A, W, H, EPS, S1[:, :k])
    assert rel(S1[:, :k], U @ H.t()) < 1e-13 and float(S1[:, k:].min()) == 7.0
    S2 = torch.full((k + 1, n + 2), 7.0, dtype=torch.float64, device=dev)
    ops.kl_wtu(A, W, H, EPS, S2[:k, :n])
    assert rel(S2[:k, :n], W.t() @ U) < 1e-13 and float(S2[k:].min()) == 7.0 and float(S2[:, n:].min()) == 7.0
    # twice the same bits
    S3 = torch.empty(m, k, dtype=torch.float64, device=dev)
    ops.kl_uht(A, W, H, EPS, S3)
    assert torch.equal(S3, S1[:, :k])


@pytest.mark.parametrize("m,n,k", [(200, 150, 6), (2100, 333, 20), (4200, 131, 64), (9000, 70, 100),
                                   (20000, 200, 40), (17000, 130, 100), (16500, 77, 20)])      # (tall shapes: four row tiles per wave)
def test_f64_products_on_unaligned_views(ops, m, n, k):
    """The big products on operands that rule out the 16-byte accesses: odd pitches, bases 8 bytes off (views of larger tensors) --
    the kernels fall back to 8-byte accesses lane by lane, same sums."""
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(7 * m + n + k)
    A = torch.rand(m + 1, n + 3, dtype=torch.float64, device=dev, generator=g)[1:, 1:n + 1]
    W = torch.rand(m, k + 1, dtype=torch.float64, device=dev, generator=g)[:, 1:]
    H = torch.rand(k, n + 1, dtype=torch.float64, device=dev, generator=g)[:, 1:]
    tol = 1e-13
    assert rel(ops.aht(A, H, torch.empty(m, k, dtype=torch.float64, device=dev)), A @ H.t()) < tol
    assert rel(ops.wta(A, W, torch.empty(k, n, dtype=torch.float64, device=dev)), W.t() @ A) < tol
    U = A / (W @ H + EPS)
    assert rel(ops.kl_uht(A, W, H, EPS, torch.empty(m, k, dtype=torch.float64, device=dev)), U @ H.t()) < tol
    assert rel(ops.kl_wtu(A, W, H, EPS, torch.empty(k, n, dtype=torch.float64, device=dev)), W.t() @ U) < tol
    R = A - W @ H
    assert abs(float(ops.resid_sqnorm(A, W, H)) / float((R * R).sum()) - 1) < tol
    Ac, Hc = A.contiguous(), H.contiguous()                          # the same products on aligned operands (the 16-byte accesses)
    assert rel(ops.aht(Ac, Hc, torch.empty(m, k, dtype=torch.float64, device=dev)), A @ H.t()) < tol
    assert rel(ops.wta(Ac, W.contiguous(), torch.empty(k, n, dtype=torch.float64, device=dev)), W.t() @ A) < tol


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


@pytest.mark.parametrize("name", [c for c in F64 if "_1x1_" in c])
def test_f64_fit_and_step_match_reference_golden(name):
    from pydnmfk_amd.dist_nmf import nmf_algorithms_1D
    from pydnmfk_amd.pyDNMF import PyNMF
    meta, A, W0, H0, z = load_case(name)
    assert A.dtype == np.float64
    for itr in meta["itrs"]:
        nmf = PyNMF(A, factors=[W0, H0], params=_args(meta["k"], itr, meta["norm"], meta["W_update"], meta.get("method", "mu")))
        assert nmf._ops().name == "hip-f64" and nmf.eps == EPS
        W, H, err = nmf.fit()
        assert W.dtype == np.float64 and H.dtype == np.float64
        assert rel_fro(W, z["r0_fit%d_W" % itr]) <= 1e-8 and rel_fro(H, z["r0_fit%d_H" % itr]) <= 1e-8, itr
        assert abs(err - float(z["r0_fit%d_err" % itr])) <= 1e-9, itr
    args = _args(meta["k"], 1, meta["norm"], meta["W_update"], meta.get("method", "mu"))
    args.m, args.n, args.eps = meta["m"], meta["n"], EPS
    W, H = torch.from_numpy(W0).cuda(), torch.from_numpy(H0).cuda()
    nmf_algorithms_1D(torch.from_numpy(A).cuda(), W, H, params=args).update()
    assert rel_fro(W.cpu().numpy(), z["r0_step1_W"]) <= 1e-10 and rel_fro(H.cpu().numpy(), z["r0_step1_H"]) <= 1e-10


@pytest.mark.parametrize("name", [c for c in F64 if "_1x1_" not in c])
def test_f64_grids_match_reference_golden(name):
    """the float64 goldens on 2 x 1, 1 x 2 and 2 x 2 grids: one process per rank on the one GPU, exchanges over gloo, the
    choreography of pydnmfk_amd/dist_nmf.py over the float64 operator set"""
    from tests._mp import run_case
    run_case(name, use_hip=True, timeout=400, tols=(1e-10, 1e-8, 1e-9))


def _ref_recipe_rank(rank, world, port, grid, q):
    """one rank of the reference's own test (tests/test_dist_nmf_1d.py:12-46): float64 data of exact rank 2, grids [1, 2] / [2, 1],
    methods mu (fro, kl) and hals, 2000 iterations from a rand init; its assertion: rel_error < 1e-3"""
    import os
    import traceback
    try:
        import torch.distributed as dist
        from pydnmfk_amd.dist_comm import MPI_comm
        from pydnmfk_amd.pyDNMF import PyNMF
        from pydnmfk_amd.utils import determine_block_params, parse
        torch.cuda.set_device(0)
        if world > 1:
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        np.random.seed(100)
        m, k, n = 24, 2, 12
        W, H = np.random.rand(m, k), np.random.rand(k, n)
        A = W @ H
        p_r, p_c = grid
        comms = MPI_comm(None, p_r, p_c)
        out = {}
        for mthd, norm in (("mu", "fro"), ("mu", "kl"), ("hals", "fro")):
            args = parse()
            args.size, args.rank, args.comm1, args.comm, args.p_r, args.p_c = world, rank, comms.comm, comms, p_r, p_c
            args.m, args.n, args.k, args.itr, args.init, args.verbose = m, n, k, 2000, "rand", False
            args.row_comm, args.col_comm = comms.cart_1d_row(), comms.cart_1d_column()
            args.method, args.norm = mthd, norm
            s, e = determine_block_params(rank, (p_r, p_c), A.shape).determine_block_index_range_asymm()
            A_ij = A[s[0]:e[0] + 1, s[1]:e[1] + 1]
            nmf = PyNMF(A_ij, factors=None, params=args)
            Wf, Hf, err = nmf.fit()
            out[(mthd, norm)] = (float(err), str(Wf.dtype), nmf._ops().name)
        q.put((rank, out, None))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        q.put((rank, None, traceback.format_exc()))


@pytest.mark.parametrize("grid", [(1, 1), (1, 2), (2, 1)])
def test_reference_own_test_recipe_in_float64(grid):
    import torch.multiprocessing as mp
    from tests._mp import free_port
    world = grid[0] * grid[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_ref_recipe_rank, args=(r, world, port, grid, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, out, err in res:
        assert err is None, "rank %d failed:\n%s" % (rank, err)
        for key, (e, dt, name) in out.items():
            assert e < 1e-3 and dt == "float64" and name == "hip-f64", (grid, rank, key, e, dt, name)


@pytest.mark.parametrize("rng", ["device", "numpy"])
def test_cli_precision_float64(tmp_path, golden_dir, rng):
    """main.py --precision float64 (reference main.py:29: the file is cast to that dtype and factorised in it): the factors come
    back float64 and reproduce the matrix as well as the reported error says -- and the run agrees with the float32 run to float32"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    A = np.load(golden_dir + "/data_swim.npz")["A"].astype(np.float64)
    np.save(tmp_path / "swimcopy.npy", A)
    errs = {}
    for prec in ("float64", "float32"):
        cmd = [sys.executable, os.path.join(root, "main.py"), "--process=pyDNMF", "--p_r=1", "--p_c=1", "--fpath=%s/" % tmp_path,
               "--fname=swimcopy", "--ftype=npy", "--k=4", "--itr=60", "--norm=fro", "--method=mu", "--precision=%s" % prec,
               "--rng=%s" % rng, "--results_path=%s/res_%s/" % (tmp_path, prec)]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        errs[prec] = float(out.stdout.strip().split("relative error =")[-1])
        W = np.load(tmp_path / ("res_%s" % prec) / "W_factors" / "W_0.npy")
        H = np.load(tmp_path / ("res_%s" % prec) / "H_factors" / "H_0.npy")
        assert W.shape == (1024, 4) and H.shape == (4, 256) and W.dtype == np.dtype(prec) and H.dtype == np.dtype(prec)
        assert abs(np.linalg.norm(A - W @ H) / np.linalg.norm(A) - errs[prec]) < (1e-10 if prec == "float64" else 1e-5)
    assert 0.3 < errs["float64"] < 0.9


def test_nmfk_sweep_in_float64(tmp_path, golden_dir):
    """the NMFk driver on float64 data (perturbations, fits, clustering, regression fit and column errors all in float64): the
    three-feature problem of the reference's NMFk fixture comes back with the same estimate and stable-cluster silhouettes"""
    from pydnmfk_amd.dist_comm import MPI_comm
    from pydnmfk_amd.pyDNMFk import PyNMFk
    from tests.test_nmfk_cpu import _args
    z = np.load(golden_dir + "/nmfk_1x1.npz")
    A = z["A"].astype(np.float64)
    comms = MPI_comm(None, 1, 1)
    nmfk = PyNMFk(A, factors=None, params=_args(tmp_path, comms))
    nopt = nmfk.fit()
    assert nopt == int(z["nopt"]) == 3
    for k in (1, 2, 3):
        st = nmfk.stats[k]
        assert np.asarray(st["L_err"]).dtype == np.float64
        assert np.allclose(st["clusterSilhouetteCoefficients"], z["k%d_clusterSilhouetteCoefficients" % k], atol=0.08), k
        assert abs(st["avgErr"] / float(z["k%d_avgErr" % k]) - 1) < 2e-2, k      # (a float64 trajectory against the fixture's float32 one)


@pytest.mark.parametrize("m,n,k,method,norm,itr", [(300, 200, 7, "mu", "fro", 23), (300, 200, 7, "mu", "kl", 23), (300, 200, 7, "hals", "fro", 12),
                                                   (1024, 256, 16, "mu", "kl", 11), (2100, 333, 40, "mu", "fro", 11), (513, 129, 3, "hals", "fro", 21)])
def test_f64_whole_fit_equals_step_loop(m, n, k, method, norm, itr):
    """dnmf_f64_fit enqueues the float64 primitives in the order the choreography issues them: the factors of a whole-fit call equal
    the Python step loop's (`params.fit_loop = 'python'`) BIT FOR BIT, with and without W updates; a batch of fits equals single fits."""
    from pydnmfk_amd.pyDNMF import PyNMF
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(m + n + k)
    A = torch.rand(m, n, dtype=torch.float64, device=dev, generator=g)
    A[:, ::6] = 0.0
    W0 = torch.rand(m, k, dtype=torch.float64, device=dev, generator=g)
    H0 = torch.rand(k, n, dtype=torch.float64, device=dev, generator=g)
    for w_update in (True, False):
        a1, a2 = _args(k, itr, norm, W_update=w_update, method=method), _args(k, itr, norm, W_update=w_update, method=method)
        a2.fit_loop = "python"
        f1, f2 = PyNMF(A, factors=[W0, H0], params=a1), PyNMF(A, factors=[W0, H0], params=a2)
        assert f1._whole_fit_ok(f1._ops()) and not f2._whole_fit_ok(f2._ops())
        W1, H1, e1 = f1.fit()
        W2, H2, e2 = f2.fit()
        assert W1.dtype == torch.float64 and torch.equal(W1, W2) and torch.equal(H1, H2), (w_update, float((W1 - W2).abs().max()))
        assert abs(e1 - e2) <= 1e-14 * max(1.0, abs(e2))
    fits = [PyNMF(A * (1.0 + 0.01 * b), factors=[W0, H0], params=_args(k, itr, norm, method=method)) for b in range(3)]
    single = [PyNMF(A * (1.0 + 0.01 * b), factors=[W0, H0], params=_args(k, itr, norm, method=method)).fit() for b in range(3)]
    for o, r in zip(PyNMF.fit_batch(fits), single):
        assert torch.equal(o[0], r[0]) and torch.equal(o[1], r[1]) and o[2] == r[2]


@pytest.mark.parametrize("m,n,k", [(24, 12, 2), (96, 21, 4), (100, 60, 16), (1, 9, 1), (300, 20, 5)])
@pytest.mark.parametrize("method,norm", [("mu", "fro"), ("mu", "kl"), ("hals", "fro")])
def test_f64_tiny_fit_equals_the_step_loop(m, n, k, method, norm):
    """Tiny float64 problems (the reference's own test sizes) run their whole fit as ONE single-workgroup launch (csrc/dnmf_f64_tiny.hip:
    A, W, H in LDS, plain FMA chains in index order); the step loop over the float64 primitives sums through MFMAs.  Same update rules
    (dist_nmf.py:716-751, :806-849, :873-934): the factors agree to 1e-11 of the largest entry after 60 steps (HALS: 1e-12 after 2 sweeps), with and without W updates, and both agree with the checker's loop in float64."""
    from oracle import nmf_oracle as orc
    from pydnmfk_amd._lib import lib
    from pydnmfk_amd.pyDNMF import PyNMF
    code = {("mu", "fro"): 0, ("mu", "kl"): 1, ("hals", "fro"): 2}[(method, norm)]
    assert lib.dnmf_f64_fit_tiny(m, n, k, code) == 1
    rs = np.random.RandomState(m + 3 * n + k)
    A = rs.rand(m, n)
    A[:, ::5] = 0.0
    W0, H0 = rs.rand(m, k), rs.rand(k, n)
    # (HALS on over-parameterised random data is chaotic once entries sit on the eps clamp: at 100 x 60, k = 16 the checker's loop, the step
    # loop and this kernel agree to 1e-16 for two sweeps and all three part ways at the third -- one flipped comparison -- so its sweeps
    # are compared over two; the float64 goldens hold the long HALS runs)
    itr = 2 if method == "hals" else 60
    tol = 1e-12 if method == "hals" else 1e-11
    for w_update in (True, False):
        a1, a2 = _args(k, itr, norm, W_update=w_update, method=method), _args(k, itr, norm, W_update=w_update, method=method)
        a2.fit_loop = "python"
        W1, H1, e1 = PyNMF(A, factors=[W0, H0], params=a1).fit()
        W2, H2, e2 = PyNMF(A, factors=[W0, H0], params=a2).fit()
        Wr, Hr, er = orc.fit_single(A, W0, H0, itr, norm=norm, W_update=w_update, method=method)
        for X, Y in ((W1, W2), (H1, H2), (W1, Wr), (H1, Hr)):
            assert W1.dtype == np.float64 and np.abs(X - Y).max() <= tol * np.abs(Y).max(), (w_update, np.abs(X - Y).max())
        assert abs(e1 - e2) <= 1e-10 * max(1.0, abs(e2)) and abs(e1 - er) <= 1e-9 * max(1.0, abs(er))
