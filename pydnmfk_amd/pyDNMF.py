"""Single-k distributed NMF driver on MI355X -- drop-in for reference pyDNMFk/pyDNMF.py (`PyNMF`).

    W, H, recon_err = PyNMF(A_ij, factors=None, save_factors=False, params=args).fit()

Same `params` attribute bag, same loop semantics (pyDNMF.py:138-182): `itr` update steps, clamp to
eps when i % 10 == 0 (after that step's update), and on the last step column-normalise W
(:185-194) and evaluate the Frobenius relative error (:205-218).  A_ij / factors may be numpy arrays
(copied to the current GPU; numpy arrays come back) or torch CUDA tensors (torch tensors come back).
Everything numeric runs in libdnmf_hip.so; there is no CPU path.

Differences from the reference, all deliberate and documented in DESIGN.md:
  * compute dtype follows the data, as in the reference (pyDNMF.py:68): float32 (the tuned path) or float64 (the fp64 matrix
    cores, engine.HipOpsF64: correctness first, one plain tile shape); `params.precision = 'bfloat16'` (or a
    bfloat16 tensor) keeps the data block in HBM as bf16 -- storage only, Frobenius mu / hals -- with W, H and every
    product in float32: the fit equals the float32 fit of the bf16-rounded data (BASELINE config 5);
  * method is 'mu' (fro / kl) or 'hals' (fro); 'bcd' is not provided; init is 'rand' or 'nnsvd' (1D grids);
  * `prune=True` (the reference's default when the attribute is absent) drops all-zero rows / columns before the
    iterations and scatters the factors back afterwards.  numpy callers get float64 factors back in that case, exactly
    as from the reference (its unprune scatters into np.zeros, utils.py:195,198); tensor callers keep float32 on the GPU.
"""
import numpy as np
import torch

from .dist_nmf import nmf_algorithms_1D, nmf_algorithms_2D
from .utils import data_operations, var_init


def _to_device(x, device, dtype=torch.float32, what="A_ij"):
    if isinstance(x, torch.Tensor):
        t = x
    else:
        t = torch.from_numpy(np.ascontiguousarray(x))
    # factors take the data's compute dtype on entry, as in the reference (`factors[i].astype(self.A_ij.dtype)`,
    # pyDNMF.py:92-96): NMFk hands the float64 medians / centroids of its clustering to the regression fit
    return t.to(device=device, dtype=dtype).contiguous()


def storage_dtype(A_ij, params):
    """How the data block is held in HBM: float32, or bfloat16 when `params.precision` says so ('bfloat16' / 'bf16')
    or the caller hands over a bfloat16 tensor.  bf16 is storage only -- W, H and all arithmetic stay float32."""
    prec = getattr(params, "precision", None)
    if isinstance(prec, str) and prec.lower() in ("bfloat16", "bf16") or prec is torch.bfloat16:
        return torch.bfloat16
    if isinstance(A_ij, torch.Tensor) and A_ij.dtype == torch.bfloat16:
        return torch.bfloat16
    # float64 data are factorised in float64 (the reference computes in A_ij's dtype, pyDNMF.py:68; `precision` only casts
    # the file in data_read, main.py:29) on the fp64 matrix cores (engine.HipOpsF64)
    dt = A_ij.dtype if isinstance(A_ij, torch.Tensor) else getattr(A_ij, "dtype", None)
    if dt in (torch.float64, np.dtype("float64")):
        return torch.float64
    return torch.float32


_MASKS = ("row_zero_idx_x", "col_zero_idx_x", "row_zero_idx_w", "col_zero_idx_h")
_NATIVE_FITS = (("mu", "fro"), ("mu", "kl"), ("hals", "fro"))


class PyNMF:
    """Reference pyDNMF.py:6-239 (the MU path of it)."""

    def __init__(self, A_ij, factors=None, save_factors=False, params=None, ops=None):
        self._numpy_io = not isinstance(A_ij, torch.Tensor)
        self.ops = ops
        if ops is None:
            if isinstance(A_ij, torch.Tensor) and not A_ij.is_cuda:
                raise TypeError("PyNMF: A_ij is a CPU tensor; pass a CUDA tensor or a numpy array (no CPU fallback)")
            if not torch.cuda.is_available():
                raise RuntimeError("PyNMF: no GPU visible; the MI355X engine has no CPU fallback")
            device = A_ij.device if isinstance(A_ij, torch.Tensor) else torch.device("cuda", torch.cuda.current_device())
        else:
            device = A_ij.device if isinstance(A_ij, torch.Tensor) else torch.device("cpu")
        self.device = device
        if ops is None and getattr(params, "shared_gpu", False):
            # (main.py --shared_gpu / params.shared_gpu = True) the launch-chain kernels from the start: no kernel of this process waits
            # for co-resident workgroups, so none can time out behind a co-tenant
            from ._lib import lib
            lib.dnmf_set_persistent(0)
        self.a_dtype = storage_dtype(A_ij, params)
        self.c_dtype = torch.float64 if self.a_dtype == torch.float64 else torch.float32      # factors, products, eps
        self.A_ij = _to_device(A_ij, device, self.a_dtype)
        self.params = params
        self.m_loc, self.n_loc = self.A_ij.shape
        self.init = self.params.init if getattr(self.params, "init", None) else 'rand'
        if "grid" in vars(self.params) and self.params.grid:
            self.p_r, self.p_c, self.k = self.params.grid[0], self.params.grid[1], self.params.k
            self.params.p_r, self.params.p_c = self.p_r, self.p_c
        else:
            self.p_r, self.p_c, self.k = self.params.p_r, self.params.p_c, self.params.k
        self.comm1 = self.params.comm1
        self.cart_1d_row, self.cart_1d_column, self.comm = self.params.row_comm, self.params.col_comm, self.params.comm
        self.verbose = self.params.verbose if getattr(self.params, "verbose", False) else False
        self.rank = self.comm1.rank
        self.eps = float(np.finfo(np.float64 if self.c_dtype == torch.float64 else np.float32).eps)   # pyDNMF.py:68
        self.params.eps = self.eps
        self.norm = var_init(self.params, 'norm', default='kl')     # :70
        self.method = var_init(self.params, 'method', default='mu')
        if self.a_dtype == torch.bfloat16 and str(self.norm).lower() == 'kl':
            raise TypeError("PyNMF: bfloat16 storage of A is provided for the Frobenius updates (mu / hals) only")
        self.prune = var_init(self.params, 'prune', default=True)
        self.save_factors = save_factors
        self.params.itr = var_init(self.params, 'itr', default=5000)
        self.itr = self.params.itr
        try:
            self.W_update = self.params.W_update
        except AttributeError:
            self.params.W_update = True
            self.W_update = True
        self.p = self.p_r * self.p_c
        self.topo = '2d' if (self.p_r != 1 and self.p_c != 1) else '1d'   # :83-87
        self.params.topo = self.topo
        self.data_op = data_operations(self.A_ij, self.params)      # :88 -> params.m, n, m_loc, n_loc, ...
        self.params = self.data_op.params
        if factors is not None:                                     # :90-96 (copied on entry)
            W0 = _to_device(factors[0], device, self.c_dtype, what="factors").clone()
            H0 = _to_device(factors[1], device, self.c_dtype, what="factors").clone()
        else:
            W0, H0 = self.init_factors()
        if self.topo == '1d':
            self.W_i, self.H_j = W0, H0
        else:
            self.W_ij, self.H_ij = W0, H0
        if self.prune:                                              # :99-101
            if self.topo == '1d':
                self.A_ij, self.W_i, self.H_j = self.data_op.prune_all(self.W_i, self.H_j)
            else:
                self.A_ij, self.W_ij, self.H_ij = self.data_op.prune_all(self.W_ij, self.H_ij)
            self.m_loc, self.n_loc = self.A_ij.shape
            # the keep-masks live on the (shared) params bag: this fit's own copy, for fits that are set up together and
            # finished later (fit_batch)
            self._masks = tuple(getattr(self.params, n_) for n_ in _MASKS)
            if self.topo == '2d':   # pruned slices are ragged: exchange the actual sizes once per fit
                self.params._slice_counts = (
                    [int(c) for c in self.cart_1d_column.allgather(int(self.W_ij.shape[0]))],
                    [int(c) for c in self.cart_1d_row.allgather(int(self.H_ij.shape[1]))])
        elif hasattr(self.params, "_slice_counts"):
            del self.params._slice_counts

    def init_factors(self):
        """pyDNMF.py:107-135, init='rand': uniform [0,1) from the process-global numpy RNG (so seeding numpy
        reproduces the reference's draw order), cast to float32; the replicated factor is broadcast from rank 0."""
        if self.init == 'nnsvd':                                    # pyDNMF.py:130-135
            if self.topo != '1d':
                raise Exception('NNSVD init only available for 1D topology, please try with 1d topo.')
            from .dist_svd import DistSVD
            return DistSVD(self.params, self.A_ij, ops=self._ops()).nnsvd(flag=1, verbose=0)
        if self.init != 'rand':
            raise NotImplementedError("init='%s': 'rand', 'nnsvd' or factors=... are provided" % self.init)
        if getattr(self.params, "rng", None) == "device" and self.device.type == "cuda":
            return self._init_factors_device()
        f32 = np.float64 if self.c_dtype == torch.float64 else np.float32        # (`.astype(self.A_ij.dtype)`, pyDNMF.py:113)
        if self.topo == '2d':
            W = np.random.rand(self.params.m_loc, self.k).astype(f32)
            H = np.random.rand(self.k, self.params.n_loc).astype(f32)
        elif self.p_c == 1:
            W = np.random.rand(self.m_loc, self.k).astype(f32)
            H = np.random.rand(self.k, self.n_loc).astype(f32) if self.rank == 0 else None
            H = self.comm1.bcast(H, root=0)
        else:
            H = np.random.rand(self.k, self.n_loc).astype(f32)
            W = np.random.rand(self.m_loc, self.k).astype(f32) if self.rank == 0 else None
            W = self.comm1.bcast(W, root=0)
        return _to_device(W, self.device, self.c_dtype), _to_device(H, self.device, self.c_dtype)

    def _init_factors_device(self):
        """`params.rng = 'device'` (what main.py / pyDNMFk_Runner select): the same uniform [0,1) init (pyDNMF.py:110-129),
        drawn ON the GPU -- no m x k / k x n host arrays, no PCIe copy per fit (an NMFk sweep makes 20 x K fits).  Seeded by
        `params.init_seed` (NMFk sets one per perturbation; per-rank factors add the rank) or torch's global generator; the
        replicated factor of a 1D grid still comes from rank 0.  A different random stream than numpy's: the parity
        fixtures use the default `rng = 'numpy'`."""
        dev = self.device
        seed = getattr(self.params, "init_seed", None)

        def draw(shape, salt):
            if seed is None:
                return torch.rand(shape, dtype=self.c_dtype, device=dev)
            g = torch.Generator(device=dev)
            g.manual_seed(int(seed) * 1000003 + salt)
            return torch.rand(shape, dtype=self.c_dtype, device=dev, generator=g)
        if self.topo == '2d':
            return draw((self.params.m_loc, self.k), 2 * self.rank + 1), draw((self.k, self.params.n_loc), 2 * self.rank + 2)
        if self.p_c == 1:
            W = draw((self.m_loc, self.k), 2 * self.rank + 1)
            H = self.comm1.bcast(draw((self.k, self.n_loc), 0) if self.rank == 0 else torch.empty(self.k, self.n_loc, dtype=self.c_dtype, device=dev), root=0)
        else:
            H = draw((self.k, self.n_loc), 2 * self.rank + 2)
            W = self.comm1.bcast(draw((self.m_loc, self.k), 0) if self.rank == 0 else torch.empty(self.m_loc, self.k, dtype=self.c_dtype, device=dev), root=0)
        return W.contiguous(), H.contiguous()

    def _ops(self):
        if self.ops is None:
            from .engine import ops_for
            self.ops = ops_for(self.params, self.c_dtype)
        return self.ops

    def _out(self, t):
        if not self._numpy_io:
            return t
        a = t.cpu().numpy()
        return a.astype(np.float64) if self.prune else a        # the reference's dtype after unprune (utils.py:195,198)

    def fit(self):
        """pyDNMF.py:138-182.  Returns (W, H, recon_err).

        Several kernels behind this loop need the GPU to themselves (their workgroups wait for each other: the whole fits of small
        problems, the persistent HALS W sweep, the one-pass MU/FRO step).  When one of them gives up -- another process or stream holds
        CUs -- the fit is NOT lost: the initial factors are kept until the end, the ranks agree on the time-out, every rank switches its
        process to the launch-chain kernels (engine.persistent_off: same update rules, no co-residency) and the fit runs again from them."""
        from ._lib import PersistentTimeout
        keep = self._initial_factors()
        nc = getattr(self.params, "_native_comm", None)
        if nc is not None and hasattr(nc, "fit_begin"):
            nc.fit_begin()
        try:
            return self._fit_once()
        except PersistentTimeout as ex:
            from .engine import persistent_off
            persistent_off(str(ex).split(" -- ")[0])
            self._restore_factors(keep)
            if nc is not None and hasattr(nc, "fit_begin"):
                nc.fit_begin()
            return self._fit_once()

    def _initial_factors(self):
        W, H = (self.W_ij, self.H_ij) if self.topo == '2d' else (self.W_i, self.H_j)
        return W.clone(), H.clone()

    def _restore_factors(self, keep):
        W, H = (self.W_ij, self.H_ij) if self.topo == '2d' else (self.W_i, self.H_j)
        W.copy_(keep[0])
        H.copy_(keep[1])
        for name in ("W_i", "H_j") if self.topo == '2d' else ():       # (the gathered copies relative_err made)
            if hasattr(self, name):
                delattr(self, name)

    def _fit_once(self):
        if self.method.lower() not in ('mu', 'hals'):
            raise NotImplementedError("method '%s' is not part of the MI355X engine (mu / hals)" % self.method)
        ops = self._ops()
        if self._whole_fit_ok(ops):
            # one rank: the whole loop below -- steps, clamps, normalisation, both squared norms -- is ONE library call
            # (dnmf_*_fit, csrc/dnmf_fit.hip): no Python, no ctypes call and no host synchronisation inside the fit
            sq = ops.fit(self.method, self.norm, self.A_ij, self.W_i, self.H_j, self.eps, self.W_update, self.itr,
                         column_sweep=(getattr(self.params, "hals_sweep", None) == "columns"))
            return self._finish(sq[0])
        for i in range(self.itr):
            clamp = (i % 10 == 0)                                   # :155 / :170, fused into the step
            if self.topo == '2d':
                self.W_ij, self.H_ij = nmf_algorithms_2D(self.A_ij, self.W_ij, self.H_ij, params=self.params,
                                                         ops=ops).update(clamp=clamp, more=(i < self.itr - 1))
            else:
                self.W_i, self.H_j = nmf_algorithms_1D(self.A_ij, self.W_i, self.H_j, params=self.params,
                                                       ops=ops).update(clamp=clamp)
            if i == self.itr - 1:
                if self.topo == '2d':
                    self.W_ij, self.H_ij = self.normalize_features(self.W_ij, self.H_ij)
                else:
                    self.W_i, self.H_j = self.normalize_features(self.W_i, self.H_j)
                return self._finish(None)

    def _whole_fit_ok(self, ops):
        """One rank, the product's own fp32-MFMA operator set, a method / norm pair the library has a whole-fit entry point for
        (anything else keeps the step loop, which raises the reference's messages for invalid pairs).  `params.fit_loop =
        'python'` keeps the step loop (A/B runs, tests of the per-step API)."""
        return (self.p == 1 and self.topo == '1d' and self.itr >= 1 and getattr(ops, "name", "") in ("hip", "hip-f64") and hasattr(ops, "fit")
                and not (getattr(ops, "name", "") == "hip-f64" and self.k > 128)
                and (str(self.method).lower(), str(self.norm).lower()) in _NATIVE_FITS
                and getattr(self.params, "fit_loop", None) != "python"
                and not getattr(self.params, "native_always", False))

    def _finish(self, sq):
        """The end of the last iteration (pyDNMF.py:158-166, :173-181): error, un-pruning, save.  `sq`: the device pair
        {sum (A - W H)^2, sum A^2} a whole-fit call left (None: evaluate them here)."""
        ops = self._ops()
        persistent = self.method.lower() == 'hals' or sq is not None or self.p == 1     # (paths that may have run a kernel of that kind)
        if persistent and hasattr(ops, "hals_check"):
            # a persistent kernel (the HALS W sweep; the whole-fit kernel of a small problem, csrc/dnmf_small.h; the one-pass MU/FRO step,
            # csrc/dnmf_team.h) that lost its co-residency raises here -- BEFORE the error's allreduce sees NaN factors, and on EVERY rank
            # (the word is per device: the ranks agree on it first, or the others would walk into the next collective alone); fit() then
            # runs the fit again on the launch-chain kernels
            from ._lib import PersistentTimeout
            bad = None
            try:
                ops.hals_check()
            except PersistentTimeout as ex:
                bad = ex
            nbad = int(self.params.comm1.allreduce(1 if bad is not None else 0))
            if nbad:
                raise bad if bad is not None else PersistentTimeout("a persistent kernel timed out on %d other rank(s)" % nbad)
        nc = getattr(self.params, "_native_comm", None)
        if nc is not None and getattr(nc, "direct_ready", False) and nc._get_direct_on():
            # a direct allreduce that gave up waiting for a peer has produced garbage since: fatal, on every rank together
            nbad = int(self.params.comm1.allreduce(1 if nc.direct_timed_out() else 0))
            if nbad:
                raise RuntimeError("direct allreduce: a wait timed out on %d rank(s) (a peer stalled longer than params.direct_timeout, "
                                   "default 30 s): the factors of this fit are invalid" % nbad)
        self.relative_err(sq)
        if self.verbose is True and self.rank == 0:
            print('relative error is:', self.recon_err)
        W, H = (self.W_ij, self.H_ij) if self.topo == '2d' else (self.W_i, self.H_j)
        if self.prune:                                      # :166,:180-181 (before the save here, so that the
            for n_, v in zip(_MASKS, getattr(self, "_masks", ())):
                setattr(self.params, n_, v)                 # (this fit's masks: the bag may have served another fit since)
            W, H = self.data_op.unprune_factors(W, H)       # saved blocks have the un-pruned shapes)
        if self.save_factors:
            from .data_io import data_write
            data_write(self.params).save_factors([W.cpu().numpy(), H.cpu().numpy()])
        return self._out(W), self._out(H), self.recon_err

    @staticmethod
    def fit_batch(fits):
        """Fit several independent problems TOGETHER: `fits` are PyNMF objects set up one after another (each has drawn
        its perturbation and its initial factors in the reference's order); the result is the list of their `fit()`
        results.  Same-shape problems on one rank run as ONE batched whole-fit call -- every kernel launch covers all of
        them (blockIdx.z = problem), each problem on the same kernels and operands as a fit of its own: the results are
        bit-identical to fitting them one by one (pyDNMFk.py:226-231 runs them one by one).  Anything the batched entry
        points do not cover falls back to exactly that."""
        from .engine import stack_alloc
        fits = list(fits)
        f0 = fits[0] if fits else None

        def same(f):
            return (f._whole_fit_ok(f._ops()) and f._ops() is f0._ops() and f.A_ij.shape == f0.A_ij.shape
                    and f.A_ij.dtype == f0.A_ij.dtype and f.A_ij.device == f0.A_ij.device and f.W_i.shape == f0.W_i.shape
                    and f.H_j.shape == f0.H_j.shape and (f.method, f.norm, f.itr, f.W_update, f.eps, f.k) ==
                    (f0.method, f0.norm, f0.itr, f0.W_update, f0.eps, f0.k))
        if len(fits) < 2 or not all(same(f) for f in fits):
            return [f.fit() for f in fits]
        if f0.method.lower() not in ('mu', 'hals'):
            raise NotImplementedError("method '%s' is not part of the MI355X engine (mu / hals)" % f0.method)
        A = getattr(f0, "_stack", None)
        if A is None or A.shape[0] != len(fits) or any(getattr(f, "_stack", None) is not A or f.A_ij.data_ptr() != A[b].data_ptr()
                                                       for b, f in enumerate(fits)):
            A = stack_alloc(len(fits), f0.A_ij.shape[0], f0.A_ij.shape[1], f0.A_ij.dtype, f0.A_ij.device)
            for b, f in enumerate(fits):                    # (callers that adopt_stack() as they go never get here)
                f.adopt_stack(A, b)
        W = stack_alloc(len(fits), f0.W_i.shape[0], f0.W_i.shape[1], f0.W_i.dtype, f0.W_i.device)
        H = stack_alloc(len(fits), f0.H_j.shape[0], f0.H_j.shape[1], f0.H_j.dtype, f0.H_j.device)
        for b, f in enumerate(fits):
            W[b].copy_(f.W_i)
            H[b].copy_(f.H_j)
        from ._lib import PersistentTimeout
        ops = f0._ops()
        for attempt in (0, 1):
            sq = ops.fit(f0.method, f0.norm, A, W, H, f0.eps, f0.W_update, f0.itr,
                         column_sweep=(getattr(f0.params, "hals_sweep", None) == "columns"))
            try:
                if hasattr(ops, "hals_check"):
                    ops.hals_check()
                break
            except PersistentTimeout as ex:                 # (one rank per problem here: nobody to agree with)
                if attempt:
                    raise
                from .engine import persistent_off
                persistent_off(str(ex).split(" -- ")[0])
                for b, f in enumerate(fits):                # the fits still hold their initial factors
                    W[b].copy_(f.W_i)
                    H[b].copy_(f.H_j)
        out = []
        sq_host = sq.cpu()                                  # ONE synchronisation for the whole batch
        for b, f in enumerate(fits):
            f.W_i, f.H_j = W[b], H[b]
            out.append(f._finish(sq_host[b]))
        return out

    def adopt_stack(self, stack, b):
        """Hold this fit's data block as slice `b` of `stack` ([B][m][n], the layout fit_batch hands to the library): copied
        there unless it already is that slice (NMFk perturbs straight into it), the block's own buffer is released."""
        if self.A_ij.data_ptr() != stack[b].data_ptr():
            stack[b].copy_(self.A_ij)
        self.A_ij = stack[b]
        self.data_op.ten = self.A_ij
        self._stack = stack

    def normalize_features(self, Wall, Hall):
        """pyDNMF.py:185-194: s = column sums of W (allreduced iff 2D or p_r != 1); W /= s + eps; H *= s^T."""
        ops = self._ops()
        s = ops.colsum(Wall, ops.zeros((self.k,), Wall))
        if self.topo == '2d' or self.p_r != 1:
            self.comm1.allreduce_(s)
        ops.scale_cols_div(Wall, s, self.eps)
        ops.scale_rows_mul(Hall, s)
        return Wall, Hall

    def cart_2d_collect_factors(self):
        """pyDNMF.py:197-202."""
        alg = nmf_algorithms_2D(self.A_ij, self.W_ij, self.H_ij, params=self.params, ops=self._ops())
        self.H_j = alg.gather_H()
        self.W_i = alg.gather_W()

    def relative_err(self, sq=None):
        """pyDNMF.py:205-218: ||A - W H||_F / ||A||_F; squared norms are summed over ranks, then sqrt.  `sq`: the pair of
        squared norms when a whole-fit call has already evaluated them (one rank: nothing to sum)."""
        ops = self._ops()
        if sq is None:
            if self.topo == '2d':
                self.cart_2d_collect_factors()
            sq = torch.cat([ops.resid_sqnorm(self.A_ij, self.W_i, self.H_j), ops.sqnorm(self.A_ij)])
            self.comm1.allreduce_(sq)
        num, den = (float(v) for v in sq.cpu())
        self.glob_norm_err, self.glob_norm_A = float(np.sqrt(num)), float(np.sqrt(den))
        self.recon_err = self.glob_norm_err / self.glob_norm_A
        return self.recon_err

    def column_err(self):
        """pyDNMF.py:221-239: per-column relative error sqrt(sum_i (A - W H)^2 / sum_i A^2) over the GLOBAL n columns
        (each rank fills its own column range, float64 allreduce).  Used once per k by NMFk; the per-column sums come
        from one kernel pass over A (dnmf_column_err: the residual tile stays in accumulators)."""
        from .utils import determine_block_params
        blk = determine_block_params(self.comm1, (self.p_r, self.p_c), (self.params.m, self.params.n))
        c0 = blk.determine_block_index_range_asymm()[0][1]
        ncol = blk.determine_block_shape_asymm()[1]
        if self.topo == '2d' and not hasattr(self, "W_i"):
            self.cart_2d_collect_factors()
        W, H = self.W_i, self.H_j
        if W.shape[0] != self.A_ij.shape[0] or H.shape[1] != self.A_ij.shape[1]:   # factors were un-pruned by fit()
            W = W[self.params.row_zero_idx_x] if W.shape[0] != self.A_ij.shape[0] else W
            H = H[:, self.params.col_zero_idx_x] if H.shape[1] != self.A_ij.shape[1] else H
        num, den = self._ops().column_err_sums(self.A_ij, W.contiguous(), H.contiguous())   # one pass over A, no temporaries
        col_num = torch.zeros(self.params.n, dtype=torch.float64, device=num.device)
        col_den = torch.zeros(self.params.n, dtype=torch.float64, device=num.device)
        keep = getattr(self.params, "col_zero_idx_x", None) if self.prune else None
        if keep is not None and int(keep.numel()) == ncol and num.numel() != ncol:
            col_num[c0:c0 + ncol][keep] = num           # pruned (all-zero) columns stay 0 / 0 -> nan, as in numpy
            col_den[c0:c0 + ncol][keep] = den
        else:
            col_num[c0:c0 + ncol] = num
            col_den[c0:c0 + ncol] = den
        self.comm1.allreduce_(col_num)
        self.comm1.allreduce_(col_den)
        return torch.sqrt(col_num / col_den).cpu().numpy()
