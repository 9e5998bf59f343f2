"""Object-style front end, drop-in for reference pyDNMFk/runner.py (`pyDNMFk_Runner`, :12-176): the Runner instance
itself is the `params` bag handed to data_read / PyNMF / PyNMFk (it carries `grid` and `k_range`, which those classes
look for with `"grid" in vars(params)`, pyDNMF.py:60-63).  One process per GPU; launch multi-rank runs with
torch.distributed.run (the reference: mpirun)."""
import os

import torch
import torch.distributed as dist

from .data_io import data_read
from .dist_comm import COMM_WORLD, MPI_comm
from .pyDNMF import PyNMF
from .pyDNMFk import PyNMFk


class pyDNMFk_Runner:
    def __init__(self, init="rand", itr=5000, norm="kl", method="mu", verbose=False, checkpoint=False,
                 timing_stats=False, prune=False, precision="float32", perturbations=20, noise_var=0.015,
                 sill_thr=0.6, sampling="uniform", process="pyDNMF", rng="device", exchange="auto", nmfk_split="data",
                 nmfk_batch=True, direct_allreduce=False):
        self.init, self.itr, self.norm, self.method = init, itr, norm, method
        self.verbose, self.checkpoint, self.timing_stats, self.prune = verbose, checkpoint, timing_stats, prune
        self.precision = precision
        self.perturbations, self.noise_var, self.sill_thr, self.sampling = perturbations, noise_var, sill_thr, sampling
        self.process = process
        self.rng = rng      # 'device': the block is uploaded once, random numbers are drawn on the GPU; 'numpy': the reference's host stream
        # who sequences the exchanges of a multi-rank step: 'torch' (torch.distributed between the launches), 'native' (whole steps
        # inside libdnmf_hip.so over its own RCCL communicators) or 'auto' (native for hals on grids with p_r > 1: main.py)
        if exchange not in ("auto", "torch", "native"):
            raise ValueError("exchange should be auto, torch or native")
        self._exchange_request = exchange
        # pyDNMFk over several GPUs: 'data' = the reference's block grid (X cut over the ranks), 'perturbations' = every rank holds the
        # whole X and fits its share of the perturbations (run(grid=[1, 1]) on any number of ranks; pydnmfk_amd/pyDNMFk.py)
        if nmfk_split not in ("data", "perturbations"):
            raise ValueError("nmfk_split should be data or perturbations")
        self.nmfk_split, self.nmfk_batch, self.direct_allreduce = nmfk_split, nmfk_batch, direct_allreduce
        self.fpath = self.ftype = self.fname = self.results_path = None
        self.k_range = self.step_k = None
        if self.process not in ["pyDNMFk", "pyDNMF"]:
            raise ValueError("process should be either pyDNMFk or pyDNMF")      # runner.py:75-76
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world > 1 and not dist.is_initialized():
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl")
        self.main_comm = COMM_WORLD()
        self.rank = self.main_comm.rank
        self.p_r = self.p_c = self.start_k = self.end_k = None

    def run(self, grid, fpath="data/", ftype="mat", fname="A_", results_path="results/", k_range=[1, 10], step_k=1, k=4):
        """runner.py:91-176.  Returns {"W", "H", "err"} (pyDNMF) or {"nopt"} (pyDNMFk)."""
        if len(grid) != 2 or len(k_range) != 2:
            raise ValueError("grid and k_range needs to be a list sized 2")
        self.p_r, self.p_c = grid[0], grid[1]
        ex = self._exchange_request
        if ex == "auto":
            ex = "native" if (str(self.method).lower() == "hals" and self.p_r > 1) else "torch"
        if ex == "native" and self.main_comm.size > 1:
            self.exchange = "native"
        elif hasattr(self, "exchange"):
            del self.exchange
        self.start_k, self.end_k = k_range[0], k_range[1]
        self.fpath, self.ftype, self.fname, self.results_path = fpath, ftype, fname, results_path
        self.results_paths = results_path
        self.k_range, self.step_k, self.grid, self.k = k_range, step_k, grid, k
        shared = self.process == "pyDNMFk" and self.nmfk_split == "perturbations" and self.main_comm.size > 1
        if shared:       # every rank reads the WHOLE matrix on a 1 x 1 grid of its own; PyNMFk shares the perturbations over main_comm
            if self.grid[0] * self.grid[1] != 1:
                raise ValueError("nmfk_split='perturbations' fits one-rank problems: grid=[1, 1] (the ranks share the perturbations)")
            from .dist_comm import SoloGrid
            self.comm = SoloGrid(self.main_comm.world_rank)
        else:
            self.comm = MPI_comm(self.main_comm, self.grid[0], self.grid[1])
        self.comm1 = self.comm.comm
        self.col_comm = self.comm.cart_1d_column()
        self.row_comm = self.comm.cart_1d_row()
        if self.verbose and self.rank == 0:
            print("Reading data now")
        A_ij = data_read(self).read()
        if shared:
            self.comm1 = self.main_comm
        numpy_out = True
        if self.rng == "device" and torch.cuda.is_available():
            import numpy as np
            from .pyDNMF import storage_dtype
            A_ij = torch.from_numpy(np.ascontiguousarray(A_ij)).to(device=torch.device("cuda", torch.cuda.current_device()),
                                                                   dtype=storage_dtype(A_ij, self))
        if self.verbose and self.rank == 0:
            print('Starting ' + self.process + '...')
        results = dict()
        if self.process == "pyDNMFk":
            results["nopt"] = PyNMFk(A_ij, factors=None, params=self).fit()
        else:
            W, H, err = PyNMF(A_ij, factors=None, params=self).fit()
            if numpy_out and isinstance(W, torch.Tensor):      # the reference's runner hands numpy arrays back
                W, H = W.cpu().numpy(), H.cpu().numpy()
            results["W"], results["H"], results["err"] = W, H, err
        if self.rank == 0 and self.verbose:
            print('Done ' + self.process + '.')
        return results
