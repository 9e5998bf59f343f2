"""NMFk: estimate the number of latent features k by NMF over perturbed copies of the data, custom clustering,
silhouettes and a Wilcoxon test on the regression errors -- drop-in for reference pyDNMFk/pyDNMFk.py
(`sample` :8-67, `PyNMFk` :70-299).  This is the main production CALLER of the MU hot path (20 x (k range) `PyNMF.fit`
calls); everything numeric inside `PyNMF` runs in libdnmf_hip.so, the logic here is host control.

    nopt = PyNMFk(A_ij, factors=None, params=args).fit()

Same `params` contract as the reference (`fpath`, `fname`, `start_k`/`end_k` or `k_range`, `step_k`, `perturbations`,
`noise_var`, `sampling`, `sill_thr`, `checkpoint`, `results_path`; defaults via var_init, pyDNMFk.py:143-164).
Differences: statistics are kept in memory for the p-value analysis (and written per k as the reference does; HDF5 when
h5py exists, else npz); no plots are drawn (plot_results is out of scope).  `params.hall_layout = 'aligned'` (not the
default) stacks the perturbations' H matrices as Hall[:, :, p] = H_p; the default 'reference' reproduces the reference's
vstack + C-order reshape (pyDNMFk.py:236-237), whose fibres mix perturbations -- it only seeds the regression fit.
"""
import os

import numpy as np
import torch
from scipy.stats import wilcoxon

from .data_io import data_write
from .dist_clustering import custom_clustering
from .pyDNMF import PyNMF
from .utils import Checkpoint, var_init


class sample:
    """Perturbed copy of the data (pyDNMFk.py:8-67).  'uniform': X * (1 + nv + 2 nv U[0,1)) element-wise (:42-44);
    'poisson': Poisson(X) (:47-49).  numpy input uses the process-global numpy RNG seeded with `seed`, i.e. the
    reference's exact stream (and the `PyNMF` rand init that follows continues that stream, as in the reference);
    CUDA tensors are perturbed on the device with a torch generator seeded the same way (same distribution,
    different stream)."""

    def __init__(self, data, noise_var, method, seed=None, out=None):
        self.X = data
        self.out = out          # optional destination of the perturbed copy (a slice of the stack an NMFk batch is fitted from)
        self.noise_var = noise_var
        self.seed = seed
        if self.seed is not None:
            np.random.seed(self.seed)
        self.method = method
        self.X_per = 0

    def randM(self):
        nv = self.noise_var
        if isinstance(self.X, torch.Tensor):
            if self.X.is_cuda:      # one fused pass with a counter-based generator (dnmf_perturb_uniform): 1 GB of traffic per fit at
                from .engine import HIP_OPS        # 65536 x 4096 bf16 where the torch expression below moves about 9 GB
                out = HIP_OPS.perturb_uniform(self.X, nv, 0 if self.seed is None else int(self.seed), out=self.out)
                if out is not None:
                    self.X_per = out
                    return
            g = torch.Generator(device=self.X.device)
            g.manual_seed(0 if self.seed is None else int(self.seed))
            M = torch.rand(self.X.shape, dtype=torch.float32, device=self.X.device, generator=g)
            self.X_per = (self.X * (2 * nv * M + nv + 1)).to(self.X.dtype)   # bf16 storage: perturb in fp32, round once
        else:
            M = 2 * nv * np.random.random_sample(self.X.shape).astype(self.X.dtype) + nv
            self.X_per = np.multiply(self.X, M + 1)

    def poisson(self):
        if isinstance(self.X, torch.Tensor):
            g = torch.Generator(device=self.X.device)
            g.manual_seed(0 if self.seed is None else int(self.seed))
            self.X_per = torch.poisson(self.X.float(), generator=g).to(self.X.dtype)
        else:
            self.X_per = np.random.poisson(self.X).astype(self.X.dtype)

    def fit(self):
        if self.method == 'uniform':
            self.randM()
        elif self.method == 'poisson':
            self.poisson()
        return self.X_per


def _np(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


class PyNMFk:
    """pyDNMFk.py:70-299."""

    def __init__(self, A_ij, factors=None, params=None, ops=None):
        self.A_ij = A_ij
        self.ops = ops
        self.local_m, self.local_n = self.A_ij.shape
        # `params.nmfk_split`: how a job of N ranks shares an NMFk sweep.  'data' (default) = the reference: X is cut into the
        # blocks of a p_r x p_c grid and every fit runs on all ranks (pyDNMFk.py:218-258).  'perturbations' = every rank
        # holds the WHOLE X (288 GB of HBM per MI355X) and fits the perturbations p = rank, rank + N, ... as one-rank problems
        # -- no exchange inside a fit; the factors then meet on every rank for the clustering (dist_clustering.py:84-160), the
        # regression fit and the statistics, which every rank evaluates on the full stacks: the numbers are those of the
        # 1 x 1 run, N times sooner through the fits.
        self.split = var_init(params, 'nmfk_split', default='data')
        if self.split not in ('data', 'perturbations'):
            raise ValueError("params.nmfk_split = %r: 'data' or 'perturbations'" % (self.split,))
        self.world = params.comm1
        if self.split == 'perturbations' and self.world.size > 1:
            import copy
            from .dist_comm import SoloGrid
            solo = SoloGrid(self.world.world_rank)
            params = copy.copy(params)          # the fits' own bag: one-rank communicators, a 1 x 1 grid
            params.comm1, params.comm, params.row_comm, params.col_comm = solo.comm, solo, solo.cart_1d_row(), solo.cart_1d_column()
            params.p_r = params.p_c = 1
            if "grid" in vars(params) and params.grid:
                params.grid = [1, 1]
            for stale in ("_native_comm", "exchange"):
                if hasattr(params, stale):
                    delattr(params, stale)
        self.params = params
        self.comm1 = self.params.comm1
        self.rank = self.world.rank
        if "grid" in vars(self.params) and self.params.grid:
            self.p_r, self.p_c = self.params.grid[0], self.params.grid[1]
        else:
            self.p_r, self.p_c = self.params.p_r, self.params.p_c
        self.fname = self.params.fname
        self.p = self.p_r * self.p_c
        self.topo = '2d' if (self.p_r != 1 and self.p_c != 1) else '1d'
        self.sampling = var_init(self.params, 'sampling', default='uniform')
        self.perturbations = var_init(self.params, 'perturbations', default=20)
        self.noise_var = var_init(self.params, 'noise_var', default=.03)
        self.step_k = var_init(self.params, 'step_k', default=1)
        if "k_range" in vars(self.params) and getattr(self.params, "grid", None):
            self.start_k, self.end_k = self.params.k_range[0], self.params.k_range[1]
        else:
            self.start_k, self.end_k = self.params.start_k, self.params.end_k
        self.first_k = self.start_k
        # how the P factor matrices H are stacked for the clustering / median (see pynmfk_per_k): 'reference' (default)
        # or 'aligned' (the layout the reference's comments describe)
        self.hall_layout = var_init(self.params, 'hall_layout', default='reference')
        self.sill_thr = var_init(params, 'sill_thr', default=0.9)
        self.verbose = var_init(params, 'verbose', default=False)
        self.params.checkpoint = var_init(params, 'checkpoint', default=True)
        self.params.rank = self.rank
        self.params.flag = 0   # 1: all perturbations factorised, 2: clustered, 3: results saved (pyDNMFk.py:165)
        self.cp = Checkpoint(checkpoint_save=self.params.checkpoint, params=self.params)
        self.stats = {}        # k -> cluster statistics (also written to disk per k)

    def fit(self):
        """pyDNMFk.py:169-215.  Returns the estimated number of latent features (same on every rank)."""
        self.params.results_path = self.params.results_path + self.params.fname + '/'
        if self.rank == 0:
            os.makedirs(self.params.results_path, exist_ok=True)
        if self.params.checkpoint:
            try:
                self.cp.load_from_checkpoint()
                self.start_k = self.cp.k + self.step_k if self.cp.flag > 3 else self.cp.k     # :191-194
            except (OSError, EOFError, AttributeError):
                pass
        for self.k in range(self.start_k, self.end_k + 1, self.step_k):
            self.params.k = self.k
            self.pynmfk_per_k()
        if self.rank == 0:
            nopt, _ = self.pvalueAnalysis()
            print('Rank estimated by NMFk = ', nopt)
        else:
            nopt = None
        nopt = self.world.bcast(nopt, root=0)
        self.world.barrier()
        from .dist_nmf import release_buffers
        release_buffers()                       # the sweep's scratch (sized for the largest k) is not kept alive
        return nopt

    def pynmfk_per_k(self):
        """pyDNMFk.py:218-258."""
        self.params.results_paths = self.params.results_path + str(self.k) + '/'
        if self.rank == 0:
            os.makedirs(self.params.results_paths, exist_ok=True)
        results = []
        if self.rank == 0:
            print('*************Computing for k=', self.k, '************')
        perturbation = 0
        # The perturbations are independent fits of one shape (:226-231 runs them one after another).  On one rank they are
        # set up in the reference's order -- perturbation p draws its data and then its initial factors from the stream
        # seeded p * 1000 -- and then fitted TOGETHER, `nb` at a time: one batched whole-fit call in which every kernel launch
        # covers all of them (PyNMF.fit_batch; bit-identical to the one-by-one fits, which nb = 1 still runs).
        from .engine import stack_alloc
        nb = self._batch_size()
        shared = self.split == 'perturbations' and self.world.size > 1
        mine = list(range(self.world.rank, self.perturbations, self.world.size)) if shared else list(range(self.perturbations))
        for p0 in range(0, len(mine), nb):
            fits, stack = [], None
            chunk = mine[p0:p0 + nb]
            for b, perturbation in enumerate(chunk):
                if self.rank == 0 and self.verbose:
                    print('Current perturbation =', perturbation)
                if nb > 1 and stack is None and isinstance(self.A_ij, torch.Tensor) and self.A_ij.is_cuda and self.A_ij.dim() == 2:
                    stack = stack_alloc(len(chunk), self.A_ij.shape[0], self.A_ij.shape[1], self.A_ij.dtype,
                                        self.A_ij.device)            # the perturbed copies are written straight into it
                data = sample(data=self.A_ij, noise_var=self.noise_var, method=self.sampling, seed=perturbation * 1000,
                              out=None if stack is None else stack[b]).fit()
                self.params.W_update = True
                if getattr(self.params, "rng", None) == "device":       # device-drawn init: one seed per (perturbation, k)
                    self.params.init_seed = perturbation * 1000 + self.k
                f = PyNMF(data, factors=None, params=self.params, ops=self.ops)                          # :230
                if nb > 1:
                    if stack is None or tuple(stack.shape[1:]) != tuple(f.A_ij.shape) or stack.dtype != f.A_ij.dtype:
                        if b == 0:           # (host data, bf16 storage of fp32 input, pruned shapes: the stack takes the fit's block)
                            stack = stack_alloc(len(chunk), f.A_ij.shape[0], f.A_ij.shape[1], f.A_ij.dtype, f.A_ij.device)
                    if stack is not None and tuple(stack.shape[1:]) == tuple(f.A_ij.shape) and stack.dtype == f.A_ij.dtype:
                        f.adopt_stack(stack, b)
                fits.append(f)
                del data
            results.extend(PyNMF.fit_batch(fits))
            for perturbation in chunk:
                self.cp._save_checkpoint(self.params.flag, perturbation, self.k)
            del fits, stack
        if shared:
            results = self._gather_fits(results, mine)
            perturbation = self.perturbations - 1
        self.params.flag = 1
        self.cp._save_checkpoint(self.params.flag, perturbation, self.k)
        # stack: W m_loc x k x P (column-major over (k, P) as the reference's order='F' reshape, :234-235), H k x n_loc x P
        # numpy in / numpy out fits: the stacks still go to the GPU for the clustering (its contractions over the row index
        # run on the update engine's kernels); with an injected checker back end they stay where they are
        dev = torch.device("cuda", torch.cuda.current_device()) if (self.ops is None and torch.cuda.is_available()) else None

        f64 = (self.A_ij.dtype == torch.float64) if isinstance(self.A_ij, torch.Tensor) else (self.A_ij.dtype == np.float64)

        def _t(x):
            t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
            # (float32 data with prune=True hand float64 factors back, as the reference does: the stacks stay float32 then;
            #  float64 data are clustered in float64)
            t = t.to(torch.float32) if (t.dtype == torch.float64 and not f64) else t
            return t.to(dev) if (dev is not None and not t.is_cuda) else t
        Ws = [_t(r[0]) for r in results]
        Hs = [_t(r[1]) for r in results]
        self.Wall = torch.stack(Ws, dim=-1)
        if self.hall_layout == 'reference':
            # exactly the reference's array: np.vstack(H_0 .. H_{P-1}) (P k x n) re-read in C order as (k, n, P), :236-237.
            # That is NOT "H of perturbation p in [:, :, p]" (entries of different perturbations, features and columns
            # share a fibre), but it is what its clustering permutes and what its median -- the initial H of the
            # regression fit below -- is taken over, so the default reproduces it.
            self.Hall = torch.cat(Hs, dim=0).reshape(self.k, Hs[0].shape[1], len(Hs)).contiguous()
        else:                                              # 'aligned': Hall[:, :, p] = H of perturbation p
            self.Hall = torch.stack(Hs, dim=-1)
        self.recon_err = [float(r[2]) for r in results]
        centroids, _, self.Hall, self.clusterSilhouetteCoefficients, self.avgSilhouetteCoefficients, _ = \
            custom_clustering(self.Wall, self.Hall, self.params, ops=self.ops).fit()                    # :239-240
        self.params.flag = 2
        self.cp._save_checkpoint(self.params.flag, perturbation, self.k)
        self.AvgH = _median_np_semantics(self.Hall)                                                     # :243
        self.AvgW = centroids
        self.params.W_update = False                                                                    # :245
        self.params.init_seed = None
        numpy_io = not isinstance(self.A_ij, torch.Tensor)
        f0 = [_np(self.AvgW), _np(self.AvgH)] if numpy_io else [self.AvgW, self.AvgH]
        regressH = PyNMF(self.A_ij, factors=f0, params=self.params, ops=self.ops)
        self.AvgW, self.AvgH, self.L_errDist = regressH.fit()                                           # :246-247
        self.col_err = regressH.column_err()                                                            # :248
        self.avgErr = float(np.mean(self.recon_err))
        mn = self.params.m * self.params.n
        self.AIC = 2 * self.k + mn * np.log(self.avgErr / mn)                                           # :250
        cluster_stats = {'clusterSilhouetteCoefficients': self.clusterSilhouetteCoefficients,
                         'avgSilhouetteCoefficients': self.avgSilhouetteCoefficients, 'L_errDist': self.L_errDist,
                         'L_err': self.col_err, 'avgErr': self.avgErr, 'recon_err': self.recon_err, 'AIC': self.AIC}
        self.stats[self.k] = cluster_stats
        if getattr(self.params, "ftype", None) is None:
            self.params.ftype = None
        if not shared or self.rank == 0:        # (every rank of a perturbation-shared sweep holds the same results: rank 0 writes)
            writer = data_write(self.params)
            writer.save_factors([_np(self.AvgW), _np(self.AvgH)], reg=True)                             # :254-256
            writer.save_cluster_results(cluster_stats)
        self.params.flag = 3
        self.cp._save_checkpoint(self.params.flag, perturbation, self.k)

    def _gather_fits(self, results, mine):
        """Perturbation-shared sweep: every rank has fitted its own perturbations; all ranks end with all (W, H, err) in
        perturbation order.  One allgather of the stacked factors per k (ranks with one perturbation fewer pad their stack)."""
        world, P = self.world, self.perturbations
        numpy_io = bool(results) and not isinstance(results[0][0], torch.Tensor)
        dev = torch.device("cuda", torch.cuda.current_device()) if (self.ops is None and torch.cuda.is_available()) else torch.device("cpu")

        def _t(x):
            t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
            return t.to(dev)
        per = -(-P // world.size)                          # perturbations of the busiest rank
        if results:
            W0, H0 = _t(results[0][0]), _t(results[0][1])
            shp = (tuple(W0.shape), tuple(H0.shape), W0.dtype)
        else:
            shp = None
        shp = [s_ for s_ in world.allgather(shp) if s_ is not None][0]
        Ws = torch.zeros((per,) + shp[0], dtype=shp[2], device=dev)
        Hs = torch.zeros((per,) + shp[1], dtype=shp[2], device=dev)
        errs = torch.zeros(per, dtype=torch.float64, device=dev)
        for i, r in enumerate(results):
            Ws[i], Hs[i], errs[i] = _t(r[0]), _t(r[1]), float(r[2])
        allW = world.allgather_blocks(Ws, [tuple(Ws.shape)] * world.size)
        allH = world.allgather_blocks(Hs, [tuple(Hs.shape)] * world.size)
        allE = world.allgather_blocks(errs, [tuple(errs.shape)] * world.size)
        out = []
        for p in range(P):
            r, i = p % world.size, p // world.size
            W, H, e = allW[r][i], allH[r][i], float(allE[r][i])
            out.append((W.cpu().numpy(), H.cpu().numpy(), e) if numpy_io else (W, H, e))
        return out

    def _batch_size(self):
        """How many perturbation fits run together (PyNMF.fit_batch): all of them on one rank with the product's own
        operators -- as many as fit next to each other in the GPU's free memory (each holds its perturbed copy of the data) --
        else 1.  `params.nmfk_batch` = False / 0 / 1 keeps the one-by-one fits, an integer caps the batch."""
        want = getattr(self.params, "nmfk_batch", True)
        if want is False or self.p != 1 or self.ops is not None or not torch.cuda.is_available():
            return 1
        cap = self.perturbations if want is True else max(1, int(want))
        try:
            free = torch.cuda.mem_get_info()[0]
        except Exception:  # noqa: BLE001
            return 1
        m, n = self.A_ij.shape
        # per problem: its perturbed copy of the data in the data's own element size (float64 data are 8 bytes; fp32 data that the fit stores as
        # bf16 still pass through a full-size copy first), the float64 operator set's m x n quotient image (one per call, counted per problem to
        # stay on the safe side), the factors and their workspaces
        esz = self.A_ij.element_size() if isinstance(self.A_ij, torch.Tensor) else int(np.dtype(getattr(self.A_ij, "dtype", np.float32)).itemsize)
        per = m * n * esz * (2.25 if esz == 8 else 1.25) + (m + n) * max(int(self.end_k), 1) * esz * 8 + (64 << 20)
        return int(max(1, min(cap, self.perturbations, (0.6 * free) // per)))

    def pvalueAnalysis(self):
        """pyDNMFk.py:261-299: walk k upwards; whenever the PREVIOUS k clustered well (min silhouette > sill_thr) and
        this k's column-error distribution differs from the current best (Wilcoxon p < 0.05), this k becomes the
        estimate."""
        from .data_io import read_cluster_results
        ks = list(range(self.first_k, self.end_k + 1, self.step_k))
        pvalue = np.ones(len(ks))
        sill_min, err = [], []
        for k in ks:
            st = self.stats.get(k)
            if st is None:                       # resumed run: statistics of finished k's come from disk
                st = read_cluster_results(self.params.results_path + str(k) + '/')
                st = {'L_err': st['L_err'], 'clusterSilhouetteCoefficients': st['clusterSilhouetteCoefficients']}
            err.append(np.asarray(st['L_err']))
            sill_min.append(round(float(np.min(np.asarray(st['clusterSilhouetteCoefficients']))), 2))
        one = err[0]
        nopt = 1
        i = 1
        while i < len(ks):
            if sill_min[i - 1] > self.sill_thr:
                pvalue[i] = wilcoxon(one, err[i])[1]
                if pvalue[i] < 0.05:
                    nopt = i
                    one = np.copy(err[i])
            i += 1
        return ks[nopt - 1], pvalue


def _median_np_semantics(t):
    """np.median over the last axis (pyDNMFk.py:243) for a torch tensor."""
    P = t.shape[-1]
    s = torch.sort(t, dim=-1).values
    if P % 2:
        return s[..., P // 2].contiguous()
    return (0.5 * (s[..., P // 2 - 1] + s[..., P // 2])).contiguous()
