"""Whole fits and batched fits (csrc/dnmf_fit.hip, include/dnmf.h "Whole fits on one rank").

  * one library call (`dnmf_*_fit`) against the per-step Python loop it replaces (`params.fit_loop = 'python'`): the same
    kernels in the same order on the same operands -- BIT-identical factors; the two squared norms come from fp64 atomic
    sums whose order is free, so the error is compared to 1e-12 relative.  SMALL MU/KL problems are the exception: their
    whole loop is one persistent kernel (csrc/dnmf_small.h) with the same update rule in another association of the fp32
    sums -- both it and the step loop are held to the checker's loop (oracle.fit_single) run in float64 (2e-4 of the largest entry);
  * a batch of B problems in one call (blockIdx.z = problem) against B calls of their own: bit-identical factors, problem
    by problem -- the property the NMFk sweep relies on when it fits its perturbations together;
  * the NMFk driver with and without batching: identical statistics and the same estimate.
The parity of the fits themselves against the reference goldens is tests/test_gpu_parity.py (which now runs through the
whole-fit entry points).
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

CASES = [  # m, n, k, method, norm, precision, itr
    (1000, 250, 9, "mu", "kl", "float32", 31),        # persistent small-fit kernel: ragged slab, ragged columns, 128-row slabs
    (4100, 400, 20, "mu", "kl", "float32", 12),       # ... 64-row slabs, k padded to 32, more slabs than fit one launch with 5 problems
    (70, 33, 3, "mu", "kl", "float32", 23),           # ... one or two slabs
    (1000, 250, 9, "mu", "fro", "float32", 31),       # the Frobenius twin of the persistent kernel
    (4100, 400, 20, "mu", "fro", "float32", 12),
    (96, 21, 4, "mu", "fro", "float32", 40),          # the reference's wtsi example shape
    (1024, 256, 9, "hals", "fro", "bfloat16", 21),    # HALS on the persistent kernel: the small NMFk sweep's shape and storage
    (1000, 250, 20, "hals", "fro", "float32", 13),    # ... ragged, k padded to 32
    (96, 21, 4, "hals", "fro", "float32", 30),
    (1024, 256, 5, "mu", "fro", "bfloat16", 21),      # MU/FRO on bf16-stored data: the Frobenius kernel with the slab in LDS as stored
    (1000, 250, 20, "mu", "fro", "bfloat16", 13),
    (1024, 256, 16, "mu", "kl", "float32", 25),       # the reference's swim example shape, 16-wide kernels
    (1024, 256, 17, "mu", "kl", "float32", 21),       # k = 17: 32-wide kernels on zero-padded factor images
    (1024, 256, 4, "mu", "fro", "float32", 25),
    (300, 200, 7, "mu", "fro", "float32", 12),        # nothing aligned
    (300, 200, 7, "mu", "kl", "float32", 12),
    (2048, 512, 6, "hals", "fro", "bfloat16", 15),    # BASELINE config 5's method and storage
    (2048, 512, 40, "hals", "fro", "float32", 11),
    (640, 384, 64, "mu", "fro", "bfloat16", 11),
    (1536, 640, 128, "mu", "kl", "float32", 6),
    (333, 129, 3, "hals", "fro", "float32", 11),
]


def _args(k, itr, norm, method, precision, **kw):
    from pydnmfk_amd.dist_comm import MPI_comm
    from pydnmfk_amd.utils import parse
    comms = MPI_comm(None, 1, 1)
    args = parse()
    args.comm1, args.comm, args.p_r, args.p_c, args.k = comms.comm, comms, 1, 1, k
    args.row_comm, args.col_comm = comms.cart_1d_row(), comms.cart_1d_column()
    args.itr, args.init, args.verbose, args.prune = itr, "rand", False, False
    args.norm, args.method, args.W_update, args.precision = norm, method, True, precision
    for key, v in kw.items():
        setattr(args, key, v)
    return args


def _problem(m, n, k, seed, precision):
    g = torch.Generator(device="cuda").manual_seed(seed)
    A = torch.rand(m, n, device="cuda", generator=g) + 0.01
    if seed % 2:
        A[:, ::7] = 0.0                                # zeros in the data (the KL quotient's edge case)
    if precision == "bfloat16":
        A = A.to(torch.bfloat16)
    return A, torch.rand(m, k, device="cuda", generator=g), torch.rand(k, n, device="cuda", generator=g)


def _persistent(m, n, k, method, norm, precision):
    from pydnmfk_amd._lib import lib
    if method == "hals":
        return lib.dnmf_hals_fit_persistent(m, n, k) != 0
    if method == "mu" and norm == "fro" and precision == "bfloat16":     # (the bf16 slab is half the size: every fp32-eligible shape is eligible)
        return lib.dnmf_mu_fit_persistent(m, n, k) != 0
    return method == "mu" and precision == "float32" and lib.dnmf_mu_fit_persistent(m, n, k) != 0


def _checker_fit_f64(A, W, H, itr, w_update, norm, method):
    """The checker's fit loop (oracle/nmf_oracle.py: fit_single -- pyDNMF.py:151-194 with dist_nmf.py:716-751, :806-849, :873-934 --
    pinned to the reference's vectors by tests/test_oracle_golden.py) on float64 copies of the device operands, with the float32
    epsilon the kernels under test use: ONE restatement of the reference's loop holds the persistent kernels and the step loop alike."""
    from oracle import nmf_oracle as orc
    Wr, Hr, _ = orc.fit_single(A.float().double().cpu().numpy(), W.double().cpu().numpy(), H.double().cpu().numpy(), itr, norm=norm,
                               W_update=w_update, method=method, eps=1.1920929e-07)
    return torch.from_numpy(Wr).cuda(), torch.from_numpy(Hr).cuda()


def _close(X, Y, tol):
    return float((X.double() - Y.double()).abs().max()) <= tol * float(Y.double().abs().max())


@pytest.mark.parametrize("m,n,k,method,norm,precision,itr", CASES)
def test_whole_fit_equals_step_loop(m, n, k, method, norm, precision, itr):
    from pydnmfk_amd.pyDNMF import PyNMF
    A, W0, H0 = _problem(m, n, k, 3, precision)
    if _persistent(m, n, k, method, norm, precision):
        for w_update in (True, False):
            W1, H1, e1 = PyNMF(A, factors=[W0, H0], params=_args(k, itr, norm, method, precision, W_update=w_update)).fit()
            W2, H2, e2 = PyNMF(A, factors=[W0, H0], params=_args(k, itr, norm, method, precision, W_update=w_update, fit_loop="python")).fit()
            Wr, Hr = _checker_fit_f64(A, W0, H0, itr, w_update, norm, method)
            tol = 1e-3 if method == "hals" else 2e-4       # (HALS: the sequential sweeps amplify fp32 rounding; the goldens hold 2e-3)
            for X, Y in ((W1, Wr), (H1, Hr), (W2, Wr), (H2, Hr)):
                assert _close(X, Y, tol), (w_update, float((X.double() - Y).abs().max()), float(Y.abs().max()))
            assert abs(e1 - e2) <= 1e-4 * max(1e-3, abs(e2)) and np.isfinite(e1)
        return
    for w_update in (True, False):
        f1 = PyNMF(A, factors=[W0, H0], params=_args(k, itr, norm, method, precision, W_update=w_update))
        assert f1._whole_fit_ok(f1._ops())
        W1, H1, e1 = f1.fit()
        f2 = PyNMF(A, factors=[W0, H0], params=_args(k, itr, norm, method, precision, W_update=w_update, fit_loop="python"))
        assert not f2._whole_fit_ok(f2._ops())
        W2, H2, e2 = f2.fit()
        assert torch.equal(W1, W2) and torch.equal(H1, H2), (w_update, float((W1 - W2).abs().max()), float((H1 - H2).abs().max()))
        assert abs(e1 - e2) <= 1e-12 * max(1.0, abs(e2))
        assert np.isfinite(e1)


@pytest.mark.parametrize("m,n,k,method,norm,precision,itr", CASES)
def test_batched_fit_is_bit_identical_to_single_fits(m, n, k, method, norm, precision, itr):
    from pydnmfk_amd.pyDNMF import PyNMF
    B = 5
    probs = [_problem(m, n, k, 10 + b, precision) for b in range(B)]
    single = [PyNMF(A, factors=[W0, H0], params=_args(k, itr, norm, method, precision)).fit() for A, W0, H0 in probs]
    fits = [PyNMF(A, factors=[W0, H0], params=_args(k, itr, norm, method, precision)) for A, W0, H0 in probs]
    batched = PyNMF.fit_batch(fits)
    assert getattr(fits[0], "_stack", None) is not None and fits[0]._stack.shape[0] == B      # it did run as ONE batch
    for b in range(B):
        assert torch.equal(batched[b][0], single[b][0]) and torch.equal(batched[b][1], single[b][1]), b
        assert abs(batched[b][2] - single[b][2]) <= 1e-12 * max(1.0, abs(single[b][2])), b
    # the problems are different problems (a batch that mapped every z to problem 0 would pass the loop above only for b = 0)
    assert not torch.equal(batched[0][0], batched[1][0])


def test_batch_falls_back_to_single_fits_on_mixed_shapes():
    from pydnmfk_amd.pyDNMF import PyNMF
    a = _problem(512, 256, 8, 1, "float32")
    b = _problem(512, 128, 8, 2, "float32")
    fits = [PyNMF(A, factors=[W0, H0], params=_args(8, 7, "fro", "mu", "float32")) for A, W0, H0 in (a, b)]
    out = PyNMF.fit_batch(fits)
    ref = [PyNMF(A, factors=[W0, H0], params=_args(8, 7, "fro", "mu", "float32")).fit() for A, W0, H0 in (a, b)]
    for o, r in zip(out, ref):
        assert torch.equal(o[0], r[0]) and torch.equal(o[1], r[1])


def test_fit_c_api_argument_errors():
    """strides that do not span a problem / are not 16-byte multiples, overlapping operands, a workspace too small"""
    from pydnmfk_amd._lib import lib
    m, n, k, B = 256, 128, 8, 2
    A = torch.rand(B, m, n, device="cuda")
    W = torch.rand(B, m, k, device="cuda")
    H = torch.rand(B, k, n, device="cuda")
    sq = torch.zeros(B, 2, dtype=torch.float64, device="cuda")
    nb = lib.dnmf_ws_bytes_fit(m, n, k, B)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def call(a_stride=m * n, w_stride=m * k, h_stride=k * n, wsb=nb, Wp=None):
        return lib.dnmf_mu_fro_fit(A.data_ptr(), m, n, n, (Wp if Wp is not None else W).data_ptr(), k, H.data_ptr(), n, k, 1.2e-7, 1, 3,
                                   B, a_stride, w_stride, h_stride, sq.data_ptr(), ws.data_ptr(), wsb, st)
    assert call() == 0
    assert call(a_stride=m * n - 4) == -1            # does not span a problem
    assert call(w_stride=m * k + 1) == -1            # not a multiple of 16 bytes
    assert call(wsb=nb - 1) == -2
    assert call(Wp=A) == -1                          # W inside A: operands overlap
    assert b"overlap" in lib.dnmf_last_error()
    torch.cuda.synchronize()


def _nmfk(A, tmp, tag, batch, method="mu", norm="kl", **kw):
    from pydnmfk_amd.dist_comm import MPI_comm
    from pydnmfk_amd.pyDNMFk import PyNMFk
    from pydnmfk_amd.utils import parse
    comms = MPI_comm(None, 1, 1)
    args = parse()
    args.comm1, args.comm, args.p_r, args.p_c = comms.comm, comms, 1, 1
    args.row_comm, args.col_comm = comms.cart_1d_row(), comms.cart_1d_column()
    args.fpath, args.fname, args.ftype = str(tmp) + "/", "b", "npy"
    args.start_k, args.end_k, args.step_k, args.sill_thr, args.itr, args.init = 2, 5, 1, 0.8, 80, "rand"
    args.noise_var, args.verbose, args.norm, args.method, args.checkpoint = 0.03, False, norm, method, False
    args.prune, args.perturbations = True, 6
    args.results_path = str(tmp) + "/results_%s/" % tag
    args.nmfk_batch = batch
    for key, v in kw.items():
        setattr(args, key, v)
    nm = PyNMFk(A, factors=None, params=args)
    return nm, nm.fit()


@pytest.mark.parametrize("method,norm,precision,io", [("mu", "kl", "float32", "numpy"), ("hals", "fro", "bfloat16", "torch"),
                                                     ("mu", "fro", "float32", "torch")])
def test_nmfk_batched_equals_one_by_one(tmp_path, method, norm, precision, io):
    """the driver fits its perturbations together (nmfk_batch, default) or one after another: same numbers, same estimate"""
    rs = np.random.RandomState(5)
    A = (rs.rand(384, 3) @ rs.rand(3, 192) + 0.01 * rs.rand(384, 192)).astype(np.float32)
    A[5, :] = 0                                        # one all-zero row: prune=True drops it in every perturbation
    X = A if io == "numpy" else torch.from_numpy(A).cuda()
    kw = dict(precision=precision) if precision != "float32" else {}
    a, nopt_a = _nmfk(X, tmp_path, "batched", True, method, norm, **kw)
    b, nopt_b = _nmfk(X, tmp_path, "single", False, method, norm, **kw)
    c, nopt_c = _nmfk(X, tmp_path, "four", 4, method, norm, **kw)      # 6 perturbations as a batch of 4 and a batch of 2
    assert a._batch_size() == 6 and b._batch_size() == 1 and c._batch_size() == 4
    assert nopt_a == nopt_b == nopt_c
    for k in a.stats:
        for key in ("recon_err", "avgErr", "clusterSilhouetteCoefficients", "avgSilhouetteCoefficients", "L_err", "L_errDist"):
            for other in (b, c):
                np.testing.assert_allclose(np.asarray(a.stats[k][key], dtype=np.float64), np.asarray(other.stats[k][key], dtype=np.float64),
                                           rtol=1e-12, atol=0, err_msg="%s k=%d" % (key, k))


def test_persistent_fit_that_loses_its_residency_ends_and_says_so():
    """Two batched persistent fits on two streams, each wanting most of the device, the first one 2.5 s long: the workgroups of the
    second cannot all be resident while the first runs.  Whatever the dispatcher does, (a) both calls END -- a barrier that gives up
    aborts every later wait of its problem, it does not cost one patience per barrier -- and (b) the status word tells the truth: set
    (then `hals_check` raises) or clear (then the second fit's factors are the ones it computes alone).  Through the C ABI with a
    workspace per call (the operator set shares one scratch buffer per device, which two concurrent fits must not)."""
    import ctypes
    import time
    from pydnmfk_amd._lib import DnmfError, lib
    from pydnmfk_amd.engine import HIP_OPS, stack_alloc
    B, m, n, k = 20, 1024, 256, 16
    assert lib.dnmf_mu_fit_persistent(m, n, k)
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(11)

    def stacks():
        A = stack_alloc(B, m, n, torch.float32, dev); A.copy_(torch.rand(B, m, n, device=dev, generator=g) + 0.01)
        W = stack_alloc(B, m, k, torch.float32, dev); W.copy_(torch.rand(B, m, k, device=dev, generator=g))
        H = stack_alloc(B, k, n, torch.float32, dev); H.copy_(torch.rand(B, k, n, device=dev, generator=g))
        return A, W, H
    nbytes = lib.dnmf_ws_bytes_fit(m, n, k, B)

    def call(A, W, H, itr, ws, sq, stream):
        rc = lib.dnmf_mu_kl_fit(A.data_ptr(), m, n, A.stride(1), W.data_ptr(), W.stride(1), H.data_ptr(), H.stride(1), k, 1.1920929e-07, 1, itr, B,
                                A.stride(0), W.stride(0), H.stride(0), sq.data_ptr(), ws.data_ptr(), ws.numel(), stream.cuda_stream)
        assert rc == 0, lib.dnmf_last_error()
    A1, W1, H1 = stacks()
    A2, W2, H2 = stacks()
    ws = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(2)]
    sq = [torch.empty(B, 2, dtype=torch.float64, device=dev) for _ in range(2)]
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    Wr, Hr = W2.clone(), H2.clone()
    call(A2, Wr, Hr, 200, ws[1], sq[1], s2)                  # the second fit alone: its reference
    torch.cuda.synchronize()
    HIP_OPS.hals_check()                                     # nothing has timed out so far
    t0 = time.time()
    call(A1, W1, H1, 130000, ws[0], sq[0], s1)               # ~2.5 s of barriers on 160 CUs
    call(A2, W2, H2, 200, ws[1], sq[1], s2)
    torch.cuda.synchronize()
    assert time.time() - t0 < 30.0
    try:
        HIP_OPS.hals_check()
        lost = False
    except DnmfError as ex:
        lost = True
        assert "resident" in str(ex)
    if not lost:
        assert torch.equal(W2, Wr) and torch.equal(H2, Hr)
    flag = ctypes.c_int(7)
    assert lib.dnmf_hals_sweep_status(ctypes.byref(flag), None) == 0 and flag.value == 0      # cleared by the query above
    # the abort path for certain: a patience of a tenth of a microsecond -- the first barrier anybody has to wait at gives up, every later
    # wait of that problem returns at once, the call ends, the status says so, the next fit (default patience) is clean again
    assert lib.dnmf_fit_set_timeout(1e-7) == 0
    try:
        call(A2, W2, H2, 5000, ws[1], sq[1], s2)
        torch.cuda.synchronize()
    finally:
        assert lib.dnmf_fit_set_timeout(2.0) == 0
    with pytest.raises(DnmfError, match="resident"):
        HIP_OPS.hals_check()
    W3, H3 = W2.clone(), H2.clone()
    W3.copy_(Wr); H3.copy_(Hr)
    call(A2, W3, H3, 50, ws[1], sq[1], s2)
    torch.cuda.synchronize()
    HIP_OPS.hals_check()
    assert torch.isfinite(W3).all() and torch.isfinite(H3).all()
    assert lib.dnmf_fit_set_timeout(0.0) == -1
    # the same for the HALS kernel, whose column norms wait at slots of their own
    def call_hals(W, H, itr):
        rc = lib.dnmf_hals_fro_fit(A2.data_ptr(), m, n, A2.stride(1), W.data_ptr(), W.stride(1), H.data_ptr(), H.stride(1), k, 1.1920929e-07, 1, itr, 0,
                                   B, A2.stride(0), W.stride(0), H.stride(0), sq[1].data_ptr(), ws[1].data_ptr(), ws[1].numel(), s2.cuda_stream)
        assert rc == 0, lib.dnmf_last_error()
    assert lib.dnmf_hals_fit_persistent(m, n, k)
    assert lib.dnmf_fit_set_timeout(1e-7) == 0
    try:
        W3.copy_(Wr); H3.copy_(Hr)
        call_hals(W3, H3, 3000)
        torch.cuda.synchronize()
    finally:
        assert lib.dnmf_fit_set_timeout(2.0) == 0
    with pytest.raises(DnmfError, match="resident"):
        HIP_OPS.hals_check()
    W3.copy_(Wr); H3.copy_(Hr)
    call_hals(W3, H3, 30)
    torch.cuda.synchronize()
    HIP_OPS.hals_check()
    assert torch.isfinite(W3).all() and torch.isfinite(H3).all()


@pytest.mark.parametrize("m,n,k", [(17, 5, 1), (33, 300, 2), (130, 47, 16), (2050, 130, 17), (8192, 64, 32), (100, 500, 9), (1500, 16, 5), (60, 2000, 12), (40, 1100, 30)])
@pytest.mark.parametrize("norm", ["kl", "fro"])
def test_persistent_fits_on_padded_operands_through_the_c_abi(m, n, k, norm):
    """The persistent small-fit kernels behind dnmf_mu_{kl,fro}_fit with leading dimensions larger than the rows (views of wider
    arrays: every pitch odd or unaligned), a batch of 3 and both settings of w_update, against the checker's loop in float64: ragged slabs, one to
    64 slabs per problem, k = 1, n < 16, streamed and LDS-resident A."""
    from pydnmfk_amd._lib import lib
    from pydnmfk_amd.engine import HIP_OPS
    if not lib.dnmf_mu_fit_persistent(m, n, k):
        pytest.skip("shape not on the persistent path")
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(m + 7 * n + k)
    B, itr, eps = 3, 23, 1.1920929e-07
    fn = lib.dnmf_mu_kl_fit if norm == "kl" else lib.dnmf_mu_fro_fit

    def pitched(rows, cols, extra):                    # [B][rows][cols] views of wider arrays; the problem stride stays a 16-byte multiple
        full = torch.rand(B, rows + 4 - rows % 4, cols + extra, device=dev, generator=g) + 0.05
        return full, full[:, :rows, :cols]
    for w_update in (1, 0):
        Af, A = pitched(m, n, 5)
        Wf, W = pitched(m, k, 3)
        Hf, H = pitched(k, n, 1)
        A[:, :, ::3] *= (torch.rand(B, m, (n + 2) // 3, device=dev, generator=g) > 0.3)      # zeros in the data
        W0, H0 = W.clone(), H.clone()
        guard_w, guard_h = Wf.clone(), Hf.clone()
        nbytes = lib.dnmf_ws_bytes_fit(m, n, k, B)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        sq = torch.empty(B, 2, dtype=torch.float64, device=dev)
        rc = fn(A.data_ptr(), m, n, A.stride(1), W.data_ptr(), W.stride(1), H.data_ptr(), H.stride(1), k, eps, w_update, itr, B,
                A.stride(0), W.stride(0), H.stride(0), sq.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, lib.dnmf_last_error()
        torch.cuda.synchronize()
        HIP_OPS.hals_check()
        for b in range(B):
            Wr, Hr = _checker_fit_f64(A[b].contiguous(), W0[b].contiguous(), H0[b].contiguous(), itr, bool(w_update), norm, "mu")
            assert _close(W[b], Wr, 3e-4) and _close(H[b], Hr, 3e-4), (w_update, b)
            R = A[b].double() - Wr @ Hr
            assert abs(float(sq[b, 0]) / float((R * R).sum()) - 1) < 1e-3 and abs(float(sq[b, 1]) / float((A[b].double() ** 2).sum()) - 1) < 1e-5
        # nothing outside the k columns / n columns of the factor views was written
        guard_w[:, :m, :k] = W; guard_h[:, :k, :n] = H
        assert torch.equal(guard_w, Wf) and torch.equal(guard_h, Hf)


def test_batched_hals_sweep_in_several_launches_equals_single_fits():
    """A batch whose persistent W sweeps do not all fit the device at once (6 problems x 128 workgroups of 512 rows) runs the sweep
    kernel on as many problems at a time as do: the factors still equal single fits bit for bit (round 4 took the column launches for
    such a batch: different rounding than the single fits' persistent sweep, and 41 us per column at BASELINE config 5's size)."""
    from pydnmfk_amd.pyDNMF import PyNMF
    m, n, k, B, itr = 65536, 64, 8, 6, 4
    probs = [_problem(m, n, k, 40 + b, "float32") for b in range(B)]
    single = [PyNMF(A, factors=[W0, H0], params=_args(k, itr, "fro", "hals", "float32")).fit() for A, W0, H0 in probs]
    fits = [PyNMF(A, factors=[W0, H0], params=_args(k, itr, "fro", "hals", "float32")) for A, W0, H0 in probs]
    batched = PyNMF.fit_batch(fits)
    assert getattr(fits[0], "_stack", None) is not None and fits[0]._stack.shape[0] == B
    for b in range(B):
        assert torch.equal(batched[b][0], single[b][0]) and torch.equal(batched[b][1], single[b][1]), b
    assert not torch.equal(batched[0][0], batched[5][0])
